"""MLAseCalculator mirror (newtonnet/utils/ase_interface.py:18-142) driven with a minimal Atoms-like object
(ase is not installed in this image).  K1 (scripts/md17_md/md.traj) pins input/output shapes and values."""
import os

import numpy as np
import pytest
import torch

from tests import util


class FakeAtoms:
    """The four accessors format_data() uses (ase_interface.py:131-142)."""
    def __init__(self, numbers, positions, cell=None, pbc=(False, False, False)):
        self.numbers = np.asarray(numbers)
        self.positions = np.asarray(positions, dtype=np.float64)
        self.cell = np.zeros((3, 3)) if cell is None else np.asarray(cell, dtype=np.float64)
        self.pbc = np.asarray(pbc, dtype=bool)

    def __len__(self):
        return len(self.numbers)

    def get_atomic_numbers(self):
        return self.numbers

    def get_positions(self, wrap=False):
        if wrap and self.pbc.any():
            frac = np.linalg.solve(self.cell.T, self.positions.T).T
            frac[:, self.pbc] %= 1.0
            return frac @ self.cell
        return self.positions

    def get_cell(self):
        return self.cell

    def get_pbc(self):
        return self.pbc


def test_format_data_cpu():
    from newtonnet_amd.utils.ase_interface import MLAseCalculator
    calc = MLAseCalculator.__new__(MLAseCalculator)
    calc.device, calc.dtype = torch.device('cpu'), torch.float32
    a = FakeAtoms([8, 1, 1], np.random.rand(3, 3), cell=np.diag([10.0, 11.0, 12.0]), pbc=(True, True, False))
    b = FakeAtoms([6], np.zeros((1, 3)))
    z, pos, cell, batch = calc.format_data([a, b])
    assert z.dtype == torch.int64 and pos.dtype == torch.float32 and batch.tolist() == [0, 0, 0, 1]
    assert cell.shape == (2, 3, 3) and torch.all(cell[0, 2] == 0) and cell[0, 0, 0] == 10.0 and torch.all(cell[1] == 0)


@pytest.mark.gpu
def test_calculator_K1_single_and_list():
    from newtonnet_amd.utils import MLAseCalculator
    k = util.load_npz('kat_md_traj.npz')
    calc = MLAseCalculator(os.path.join(util.GOLDEN, 'ckpt_state.npz'), properties=['energy', 'forces'],
                           precision='single', device='cuda')                     # simulate.py:11-16
    frame0 = FakeAtoms(k['numbers'], k['positions'][0])
    calc.calculate(frame0)
    assert calc.results['energy'].shape == () and calc.results['forces'].shape == (21, 3)
    assert abs(float(calc.results['energy']) - (-17591.826171875)) <= 2e-3        # md.log:2
    assert np.abs(calc.results['forces'] - k['forces'][0]).max() < 5e-5
    frames = [FakeAtoms(k['numbers'], p) for p in k['positions'][:16]]
    calc.calculate(frames)
    assert calc.results['energy'].shape == (16,) and calc.results['forces'].shape == (16, 21, 3)
    assert np.all(np.abs(calc.results['energy'] - k['energy'][:16]) <= util.energy_tol(k['energy'][:16]))
    assert np.abs(calc.results['forces'] - k['forces'][:16]).max() < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize('capture', [True, False])
def test_md_loop_path_follows_K1_trajectory(capture):
    """MD-loop path (Verlet-skin list reuse + HIP-graph replay): all 201 frames of the authors' trajectory, one calculate()
    per frame like simulate.py's dynamics loop, against their stored energies / forces (K1)."""
    from newtonnet_amd.utils import MLAseCalculator
    k = util.load_npz('kat_md_traj.npz')
    calc = MLAseCalculator(os.path.join(util.GOLDEN, 'ckpt_state.npz'), properties=['energy', 'forces'],
                           precision='single', device='cuda', skin=0.5, capture=capture)
    exact = MLAseCalculator(os.path.join(util.GOLDEN, 'ckpt_state.npz'), properties=['energy', 'forces'],
                            precision='single', device='cuda', skin=0.0)
    for t in range(len(k['positions'])):
        calc.calculate(FakeAtoms(k['numbers'], k['positions'][t]))
        assert abs(float(calc.results['energy']) - k['energy'][t]) <= util.energy_tol(k['energy'][t:t + 1])[0]
        assert np.abs(calc.results['forces'] - k['forces'][t]).max() < 5e-5
        if t % 40 == 0:     # and against the exact-list path of this package: identical up to fp32 summation order
            exact.calculate(FakeAtoms(k['numbers'], k['positions'][t]))
            assert np.abs(calc.results['forces'] - exact.results['forces']).max() < 2e-6
            assert abs(float(calc.results['energy']) - float(exact.results['energy'])) <= 4e-3
    assert calc.md_stats['steps'] == 201 and (calc._md['graph'] is not None) == capture
    # The stored frames are 100 MD steps apart (every frame moves some atom by more than skin / 2: a rebuild each time).
    # Fine-grained motion -- 60 interpolated steps between consecutive frames -- must reuse the list and still agree with
    # the exact-list path at every step.
    before = calc.md_stats['rebuilds']
    for t in (0, 57):
        for sub in range(60):
            w = sub / 60.0
            a = FakeAtoms(k['numbers'], (1 - w) * k['positions'][t] + w * k['positions'][t + 1])
            calc.calculate(a)
            if sub % 6 == 0:
                exact.calculate(a)
                assert np.abs(calc.results['forces'] - exact.results['forces']).max() < 2e-6
                assert abs(float(calc.results['energy']) - float(exact.results['energy'])) <= 4e-3
    assert 2 <= calc.md_stats['rebuilds'] - before <= 30                           # 120 steps, a handful of rebuilds


@pytest.mark.gpu
@pytest.mark.parametrize('spatial_order', [False, True])
def test_md_loop_path_periodic_with_stress(spatial_order):
    """Periodic box drifting through the cell boundary: wrapped positions jump by a lattice vector while the skin criterion
    follows the unwrapped ones; energy / forces / stress must match the exact-list path at every step.  spatial_order: the MD
    path keeps a big system in Morton order of its cells between list rebuilds (forced on for this small box, atoms shuffled so
    that the order matters); forces must come back in the Atoms object's order."""
    from newtonnet_amd.models import NewtonNet
    from newtonnet_amd.utils import MLAseCalculator
    torch.manual_seed(5)
    rng = np.random.default_rng(5)
    n, box = 64, 9.5
    grid = np.stack(np.meshgrid(*[np.arange(4)] * 3, indexing='ij'), -1).reshape(-1, 3) * (box / 4)
    pos = grid + rng.normal(0, 0.15, grid.shape)
    if spatial_order:
        pos = pos[rng.permutation(n)]
    numbers = rng.choice([1, 6, 8], n)
    cell = np.diag([box] * 3)
    vel = rng.normal(0, 0.02, pos.shape) + np.array([0.05, 0.0, 0.0])     # net drift: atoms cross the boundary
    def make(skin):
        torch.manual_seed(5)
        m = NewtonNet(output_properties=['energy', 'gradient_force', 'stress'])
        calc = MLAseCalculator(m, properties=['energy', 'forces', 'stress'], device='cuda', skin=skin)
        calc.model.__dict__['_spatial_order_min'] = 1 if (spatial_order and skin > 0) else 0
        return calc
    fast, exact = make(0.6), make(0.0)
    for step in range(30):
        # NPT-like breathing of the box (positions scale with it): the list survives small strains, see _calculate_md
        lam = 1.0 + 0.004 * np.sin(0.7 * step)
        a = FakeAtoms(numbers, (pos + step * vel) * lam, cell=cell * lam, pbc=(True, True, True))
        fast.calculate(a)
        exact.calculate(a)
        assert np.abs(fast.results['forces'] - exact.results['forces']).max() < 5e-6
        assert abs(float(fast.results['energy']) - float(exact.results['energy'])) < 1e-4
        assert fast.results['stress'].shape == (6,)
        assert np.abs(fast.results['stress'] - exact.results['stress']).max() < 1e-6
    assert 1 <= fast.md_stats['rebuilds'] < 15
    assert (fast._md['order'] is not None) == spatial_order


@pytest.mark.gpu
@pytest.mark.parametrize('activation', ['swish', 'sigmoid', 'softplus', 'tanh'])
def test_skin_list_equals_exact_list_for_every_activation(activation):
    """Reused candidate lists (Verlet skin) hold pairs that are outside the cutoff at the current step.  Their message is
    zero, but phi = W2 act(W1 0) is not when act(0) != 0 (sigmoid: 0.5, softplus: ln 2) -- the force kernels mask them by the
    neighbor predicate.  skin = 0.5 must reproduce the exact list (skin = 0) for every activation."""
    from newtonnet_amd.models import NewtonNet
    from newtonnet_amd.utils import MLAseCalculator
    k = util.load_npz('kat_md_traj.npz')

    def make(skin):
        torch.manual_seed(13)
        m = NewtonNet(activation=activation, output_properties=['energy', 'gradient_force'])
        # A seeded default-initialised model with an activation that never vanishes (softplus) blows its energy up to 3e8 eV and
        # its forces to 5e9 eV/A through cancelling terms: any reordering of a row's fp32 sum then shows at several 1e-6 relative.
        # Quarter-size interaction weights keep every activation's model at a physical scale (fix the model, not the bound).
        with torch.no_grad():
            for q in m.interaction_layers.parameters():
                q.mul_(0.25)
        return MLAseCalculator(m, properties=['energy', 'forces'], device='cuda', skin=skin)
    fast, exact = make(0.5), make(0.0)
    for t in (0, 1, 2):
        for sub in range(0, 60, 12):
            w = sub / 60.0
            a = FakeAtoms(k['numbers'], (1 - w) * k['positions'][t] + w * k['positions'][t + 1])
            fast.calculate(a)
            exact.calculate(a)
            n_cand = fast._md['g'].n_edges
            fs = max(1.0, float(np.abs(exact.results['forces']).max()))
            assert np.abs(fast.results['forces'] - exact.results['forces']).max() < 2e-6 * fs, (activation, t, sub)
            # (the skin list sums a row's pairs in a different order than the exact list: fp32 reordering only)
            assert abs(float(fast.results['energy']) - float(exact.results['energy'])) <= 2e-6 * max(
                1.0, abs(float(exact.results['energy'])))
    assert n_cand > 306           # the skin list really holds candidates beyond the cutoff (exact list of frame 0: 306)


def _save_fake_reference_pickle(path, sd, old_layout=False):
    """A whole-module pickle that names classes of a `newtonnet` package (as the reference's trainer writes,
    trainer.py:219) -- built from throw-away classes registered under those module names, which are removed again before
    loading so that the loader has to work without the reference being importable."""
    import sys
    import types
    from torch import nn
    created = []

    def cls(module, name, base=nn.Module):
        if module not in sys.modules:
            sys.modules[module] = types.ModuleType(module)
            created.append(module)
        c = type(name, (base,), {'__module__': module})
        setattr(sys.modules[module], name, c)
        return c

    for pkg in ('newtonnet', 'newtonnet.models', 'newtonnet.layers'):
        if pkg not in sys.modules:
            sys.modules[pkg] = types.ModuleType(pkg)
            created.append(pkg)
    NN = cls('newtonnet.models.newtonnet', 'NewtonNet')
    Emb = cls('newtonnet.models.newtonnet', 'EmbeddingNet')
    Inter = cls('newtonnet.models.newtonnet', 'InteractionNet')
    Edge = cls('newtonnet.layers.representations', 'EdgeEmbedding')
    Norm = cls('newtonnet.layers.representations', 'ScaledNorm')
    Bessel = cls('newtonnet.layers.representations', 'RadialBesselLayer')
    EOut = cls('newtonnet.models.output', 'EnergyOutput')
    GOut = cls('newtonnet.models.output', 'GradientForceOutput')
    SS = cls('newtonnet.layers.scalers', 'ScaleShift')
    Null = cls('newtonnet.layers.scalers', 'NullScaleShift')
    F = 128

    def lin(prefix, bias=True):
        l = nn.Linear(F, sd[prefix + '.weight'].shape[0], bias=bias)
        l.weight.data = sd[prefix + '.weight'].float().clone()
        if l.weight.shape[1] != sd[prefix + '.weight'].shape[1]:
            l = nn.Linear(sd[prefix + '.weight'].shape[1], sd[prefix + '.weight'].shape[0], bias=bias)
            l.weight.data = sd[prefix + '.weight'].float().clone()
        if bias:
            l.bias.data = sd[prefix + '.bias'].float().clone()
        return l
    m = NN()
    emb = Emb()
    emb.node_embedding = nn.Embedding(119, F, padding_idx=0)
    emb.node_embedding.weight.data = sd['embedding_layers.node_embedding.weight'].float().clone()
    edge = Edge()
    edge.norm = Norm()
    edge.norm.r = 5.0
    bes = Bessel()
    bes.frequencies = nn.Parameter(sd['embedding_layers.edge_embedding.embedding.frequencies'].float().clone(),
                                   requires_grad=False)
    if old_layout:
        edge.frequencies = bes.frequencies
    else:
        edge.embedding = bes
    emb.edge_embedding = edge
    emb.requires_dr = True
    setattr(m, 'embedding_layer' if old_layout else 'embedding_layers', emb)
    layers = []
    for l in range(3):
        p = f'interaction_layers.{l}.'
        il = Inter()
        il.message_nodepart = nn.Sequential(lin(p + 'message_nodepart.0'), nn.SiLU(), lin(p + 'message_nodepart.2'))
        il.message_edgepart = lin(p + 'message_edgepart', bias=False)
        il.equiv_message1 = nn.Sequential(lin(p + 'equiv_message1.0', False), nn.SiLU(), lin(p + 'equiv_message1.2', False))
        il.equiv_message2 = nn.Sequential(lin(p + 'equiv_message2.0', False), nn.SiLU(), lin(p + 'equiv_message2.2', False))
        il.equiv_update = lin(p + 'equiv_update', False)
        layers.append(il)
    m.interaction_layers = nn.ModuleList(layers)
    m.output_properties = ['energy', 'gradient_force']
    eo = EOut()
    eo.layers = nn.Sequential(lin('output_layers.0.layers.0'), nn.SiLU(), lin('output_layers.0.layers.2'), nn.SiLU(),
                              lin('output_layers.0.layers.4'))
    m.output_layers = nn.ModuleList([eo, GOut()])
    ss = SS()
    ss.scale = nn.Embedding(119, 1)
    ss.scale.weight.data = sd['scalers.0.scale.weight'].float().clone()
    ss.shift = nn.Embedding(119, 1)
    ss.shift.weight.data = sd['scalers.0.shift.weight'].float().clone()
    m.scalers = nn.ModuleList([ss, Null()])
    torch.save(m, path)
    for name in created:
        sys.modules.pop(name, None)
    for name in [k for k in sys.modules if k == 'newtonnet' or k.startswith('newtonnet.')]:
        sys.modules.pop(name, None)


@pytest.mark.parametrize('old_layout', [False, True])
def test_load_model_accepts_reference_saved_module(tmp_path, old_layout):
    """ase_interface.py:83-129 loads whole-module pickles written by the reference's trainer.  The mirror must ingest such a
    file -- here one whose classes live in a `newtonnet` package that is NOT importable at load time, in the current and in
    the pre-2.0 (`embedding_layer.*`) parameter layout -- by rebuilding its own NewtonNet from the state_dict."""
    from newtonnet_amd.models import NewtonNet
    from newtonnet_amd.utils.ase_interface import MLAseCalculator
    sd = util.load_state('ckpt', torch.float32)
    path = str(tmp_path / 'best_model.pt')
    _save_fake_reference_pickle(path, sd, old_layout)
    import importlib.util
    assert importlib.util.find_spec('newtonnet') is None          # the reference package is not importable here
    calc = MLAseCalculator.__new__(MLAseCalculator)
    calc.device, calc.dtype, calc.properties = torch.device('cpu'), torch.float32, ['energy', 'forces']
    model = calc.load_model(path)
    assert isinstance(model, NewtonNet) and model.output_properties == ['energy', 'gradient_force']
    assert model.embedding_layers.edge_embedding.cutoff == 5.0 and model.activation_name == 'swish'
    got = model.state_dict()
    assert set(got) == set(sd)
    for k, v in sd.items():
        assert torch.equal(got[k].float(), v), k
    # a live module that is not ours (e.g. the reference's class when its package IS importable) converts the same way
    obj = torch.load(path, map_location='cpu', weights_only=False,
                     pickle_module=__import__('newtonnet_amd.utils.ase_interface', fromlist=['x'])._ReferencePickle)
    m2 = calc.load_model(obj)
    assert isinstance(m2, NewtonNet) and torch.equal(m2.state_dict()['scalers.0.shift.weight'].float(),
                                                     sd['scalers.0.shift.weight'])


@pytest.mark.gpu
def test_md_loop_conserves_total_energy():
    """A property no oracle is needed for: the forces the MD-loop path returns are the gradient of the energy it returns.  Velocity
    Verlet with unit masses on one aspirin molecule (seeded weights), 1500 steps through MLAseCalculator.calculate with the
    Verlet-skin list (several rebuilds on the way): potential + kinetic energy stays within 1e-5 eV of its start (measured 4e-7) while a tenth of
    an eV and more flows between the two."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from newtonnet_amd.models import NewtonNet
    from newtonnet_amd.utils import MLAseCalculator
    torch.manual_seed(0)
    model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
    model.eval()
    z, pos, _, _ = bench.synthetic_aspirin(1, 0, 'cpu')
    calc = MLAseCalculator(model, properties=['energy', 'forces'], device='cuda', skin=0.5)

    def evaluate(p):
        calc.calculate(FakeAtoms(z.numpy(), p))
        f, e = calc.results['forces'].astype(np.float64), float(calc.results['energy'])
        assert np.isfinite(f).all() and np.isfinite(e)
        return e, f
    p, v, dt = pos.double().numpy().copy(), np.zeros((21, 3)), 0.002
    e, f = evaluate(p)
    totals, kin = [e], [0.0]
    for _ in range(1500):
        v = v + 0.5 * dt * f
        p = p + dt * v
        e, f = evaluate(p)
        v = v + 0.5 * dt * f
        kin.append(0.5 * (v ** 2).sum())
        totals.append(e + kin[-1])
    totals = np.asarray(totals)
    print(f'MD energy conservation: drift {np.abs(totals - totals[0]).max():.2e} eV over 1500 steps, kinetic energy up to '
          f'{max(kin):.3f} eV, list rebuilds {calc.md_stats["rebuilds"]}')
    assert calc.md_stats['rebuilds'] >= 1 and max(kin) > 0.05
    assert np.abs(totals - totals[0]).max() <= 1e-5          # (measured: 4e-7)
