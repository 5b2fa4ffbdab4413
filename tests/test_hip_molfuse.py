"""GPU parity tests of the molecule-resident fused edge phase (csrc/molfuse2.hip: mol2_edge_fwd_kernel / mol2_edge_bwd_kernel --
msg, both edge MLPs, the force-message aggregation and their adjoints in one launch per layer and direction, persistent workgroups
that take whole molecules from a device-side queue, largest first; reference semantics newtonnet/models/newtonnet.py:207-227 and the
autograd sweep of output.py:66-73).

The form is chosen by the library (pipeline.hip: batches of molecules of at most NNHIP_MOL_STAGE_MAX atoms, from a molecule count
up); hip.set_mol_fused forces it on (1), off (0) or on for one direction only (2 forward, 3 adjoint) -- every array between the two
directions has the row path's layout, so each direction is also checked against the three row-path launches it replaces.
Tolerances as tests/test_hip_parity.py (fp32 path vs fp64 oracle): force MAE <= 1e-5, max <= 5e-5 eV/A, energy within 2 ulp."""

import numpy as np
import pytest
import torch

from tests import util
from tests.test_hip_parity import check_forces, make_model

pytestmark = pytest.mark.gpu


class fused_mode:
    def __init__(self, mode):
        self.mode, self.old = mode, None

    def __enter__(self):
        from newtonnet_amd import hip
        self.old = hip.set_mol_fused(self.mode)

    def __exit__(self, *exc):
        from newtonnet_amd import hip
        hip.set_mol_fused(self.old)


def fused_launches(fn):
    """Run fn() with the library's event timers on; returns (result, launches of mol2_edge_fwd, launches of mol2_edge_bwd)."""
    from newtonnet_amd import hip
    hip.timers_enable(True, classes=('mol_fwd', 'mol_bwd'))
    try:
        res = fn()
        torch.cuda.synchronize()
        tm = hip.timers_read(reset=True)
    finally:
        hip.timers_enable(False)
    return res, tm['mol_fwd'][1], tm['mol_bwd'][1]


def molecule_zoo(gen, clusters=True):
    """40 aspirin conformers; ethanol-sized and smaller fragments; single atoms and a far-apart pair (rows without edges); a
    24-atom cluster with all 276 pairs inside the cutoff (nine pair tiles: the most a molecule can have) and a 23-atom one."""
    a = util.load_npz('aspirin_frames.npz')
    base = torch.from_numpy(a['train_pos'][0]).float()
    zb = torch.from_numpy(a['z']).long()
    zs, ps = [], []
    for k in range(40):
        zs.append(zb), ps.append(base + 0.05 * torch.randn(21, 3, generator=gen))
    for n in (9, 5, 3, 2, 1, 1):
        zs.append(zb[:n]), ps.append(base[:n] + 0.05 * torch.randn(n, 3, generator=gen))
    zs.append(zb[:2]), ps.append(torch.tensor([[0.0, 0.0, 0.0], [7.5, 0.0, 0.0]]))       # beyond the cutoff: no edge
    for n in ((24, 23) if clusters else ()):
        # points on a jittered 3x3x3 lattice 1.15 A apart: diameter < 4.3 A, every pair an edge, no close contacts
        grid = torch.stack(torch.meshgrid(*[torch.arange(3)] * 3, indexing='ij'), dim=-1).reshape(-1, 3)[:n].float()
        zs.append(zb[torch.randint(0, 21, (n,), generator=gen)]), ps.append(1.15 * grid + 0.08 * torch.randn(n, 3, generator=gen))
    sizes = [len(t) for t in zs]
    ps = [p + 30.0 * k for k, p in enumerate(ps)]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    return torch.cat(zs), torch.cat(ps), torch.zeros(len(sizes), 3, 3), batch, sizes


@pytest.mark.parametrize('which,mode', [('rand', 1), ('ckpt', 1)])
def test_fused_edge_phase_against_the_oracle(which, mode):
    """Both directions fused, on a zoo of molecules of 1 .. 24 atoms (47 molecules: fewer than the persistent grid, every workgroup
    takes one; the queue itself is exercised by the tests below)."""
    from oracle import newtonnet_ref as ref
    gen = torch.Generator().manual_seed(11)
    # (the dense lattice clusters only with the random weights: the trained model explodes on them -- energies of 1e20 eV -- and
    # nothing can be said about fp32 sums of such terms)
    z, pos, cell, batch, sizes = molecule_zoo(gen, clusters=which == 'rand')
    model, sd = make_model(which)
    args = (z.cuda(), pos.cuda(), cell.cuda(), batch.cuda())
    with fused_mode(mode):
        out, n_fwd, n_bwd = fused_launches(lambda: model(*args))
        assert (n_fwd, n_bwd) == (3, 3), f'the fused kernels did not run ({n_fwd} forward, {n_bwd} adjoint launches)'
        e, f = out.energy.cpu().double().numpy(), out.gradient_force.cpu().numpy()
        a_node, f_node = out.atom_node.cpu().double(), out.force_node.cpu().double()
        again = model(*args)                      # the deferred path: same kernels, same bits
        assert torch.equal(again.energy, out.energy) and torch.equal(again.gradient_force, out.gradient_force)
    o = ref.energy_forces({k: v.double() for k, v in sd.items()}, z, pos.double(), cell.double(), batch)
    assert np.array_equal(out.edge_index.cpu().numpy(), o['edge_index'].numpy())
    pairs = np.bincount(batch.numpy()[o['edge_index'][0].numpy()], minlength=len(sizes)) // 2
    assert pairs.max() == (276 if which == 'rand' else pairs.max()) and (pairs == 0).sum() >= 3, pairs     # the nine-tile molecule and the edge-free ones are in
    f_ref = o['forces'].numpy()
    fscale = max(1.0, float(np.abs(f_ref).max()) / 5.0)
    d = np.abs(f.astype(np.float64) - f_ref)
    print(f'fused edge phase ({which}, mode {mode}): force MAE {d.mean():.2e} max {d.max():.2e} (max |F| {np.abs(f_ref).max():.1f})')
    check_forces(f, f_ref, fscale)
    e_ref = o['energy'].numpy()
    assert np.all(np.abs(e - e_ref) <= util.energy_tol(e_ref)), np.abs(e - e_ref).max()
    np.testing.assert_allclose(a_node.numpy(), o['atom_node'].numpy(), rtol=2e-4, atol=2e-5 * fscale)
    np.testing.assert_allclose(f_node.numpy(), o['force_node'].numpy(), rtol=2e-4, atol=2e-5 * fscale)
    # ... and next to the row path on the same inputs
    with fused_mode(0):
        row, n_fwd, n_bwd = fused_launches(lambda: model(*args))
        assert (n_fwd, n_bwd) == (0, 0)
    dd = (row.gradient_force - out.gradient_force).abs().max().item()
    print(f'  fused vs row path: max |dF| {dd:.2e}, max |dE| {(row.energy - out.energy).abs().max().item():.2e}')
    assert dd <= 2e-5 * fscale


@pytest.mark.parametrize('mode,B', [(1, 96), (1, 1400)])
def test_fused_edge_phase_is_deterministic_and_ignores_the_order_of_the_molecules(mode, B):
    """96 molecules: a workgroup each.  1400: the persistent grid (512 workgroups) hands 888 molecules out through the queue, in an
    order that depends on timing -- the results must not."""
    gen = torch.Generator().manual_seed(12)
    a = util.load_npz('aspirin_frames.npz')
    n = 21
    pos = torch.from_numpy(a['test0_pos']).float().repeat(B, 1) + 0.05 * torch.randn(B * n, 3, generator=gen)
    z = torch.from_numpy(a['z']).long().repeat(B).cuda()
    batch = torch.repeat_interleave(torch.arange(B), n).cuda()
    cell = torch.zeros(B, 3, 3, device='cuda')
    model, _ = make_model('rand')
    with fused_mode(mode):
        o1 = model(z, pos.cuda(), cell, batch)
        e1, f1 = o1.energy.clone(), o1.gradient_force.clone()
        o2 = model(z, pos.cuda(), cell, batch)
        assert torch.equal(o2.energy, e1) and torch.equal(o2.gradient_force, f1)
        perm = torch.randperm(B, generator=gen)
        pos_p = pos.view(B, n, 3)[perm].reshape(-1, 3).contiguous().cuda()
        o3 = model(z, pos_p, cell, batch)
        assert torch.equal(o3.energy.cpu(), e1.cpu()[perm])
        assert torch.equal(o3.gradient_force.cpu().view(B, n, 3), f1.cpu().view(B, n, 3)[perm])
    net = f1.view(B, n, 3).sum(1).abs().max().item()
    assert net < 2e-4, net


def test_the_queue_serves_a_mix_of_molecule_sizes():
    """~1200 molecules of 1 .. 24 atoms (26 copies of the zoo, each with its own noise): more than twice the persistent grid, so most
    molecules come off the queue, in the largest-first order of mol2_order_kernel.  The results must not depend on the schedule:
    repeat = same bits; the molecules in another order (another order array, another hand-out) = the same bits per molecule; and the
    row path on the same batch agrees to rounding."""
    gen = torch.Generator().manual_seed(21)
    zs, ps, sizes = [], [], []
    for _ in range(26):
        z, pos, _cell, _batch, sz = molecule_zoo(gen, clusters=True)
        off = 0
        for n in sz:
            zs.append(z[off:off + n]), ps.append(pos[off:off + n] - pos[off:off + n].mean(0, keepdim=True)), sizes.append(n)
            off += n
    B = len(sizes)
    assert B > 1100 and max(sizes) == 24 and min(sizes) == 1

    def batch_of(order):
        z = torch.cat([zs[k] for k in order]).cuda()
        pos = torch.cat([ps[k] + 40.0 * i for i, k in enumerate(order)]).cuda()
        batch = torch.repeat_interleave(torch.arange(B), torch.tensor([sizes[k] for k in order])).cuda()
        return z, pos, torch.zeros(B, 3, 3, device='cuda'), batch

    model, _ = make_model('rand')
    ident = list(range(B))
    # (positions are molecule-local + a per-slot offset: compare per molecule in the molecule's own frame -> use the forces and energies)
    with fused_mode(1):
        out, n_fwd, n_bwd = fused_launches(lambda: model(*batch_of(ident)))
        assert (n_fwd, n_bwd) == (3, 3)
        e1, f1 = out.energy.clone(), out.gradient_force.clone()
        again = model(*batch_of(ident))
        assert torch.equal(again.energy, e1) and torch.equal(again.gradient_force, f1)
    with fused_mode(0):
        row = model(*batch_of(ident))
    fscale = max(1.0, row.gradient_force.abs().max().item() / 5.0)
    assert (row.gradient_force - f1).abs().max().item() <= 2e-5 * fscale
    assert (row.energy - e1).abs().max().item() <= 2e-5 * max(1.0, row.energy.abs().max().item())
    # a permutation that keeps every molecule at its offset slot is impossible (sizes differ), so the slot offsets move with the
    # molecules: the geometry of a molecule changes by a translation of k * 40 A, which changes fp32 bits.  Compare instead two
    # orders of the SAME slots: swap molecules of equal size only.
    by_size = {}
    for k, n in enumerate(sizes):
        by_size.setdefault(n, []).append(k)
    perm = ident[:]
    for n, ks in by_size.items():
        ks = ks[1::2]                                   # (every other member of a size class moves, the rest keep their slots)
        sh = [ks[i] for i in torch.randperm(len(ks), generator=gen).tolist()]
        for a, b in zip(ks, sh):
            perm[a] = b
    with fused_mode(1):
        o3 = model(*batch_of(perm))
    starts = np.concatenate([[0], np.cumsum(sizes)])
    # slot i of the permuted batch holds molecule perm[i] at offset 40 i; in the identity batch that molecule sat at offset 40 perm[i].
    # Only molecules whose slot did not move can be compared bit for bit; for the others compare to rounding.
    same = moved = 0
    for i, k in enumerate(perm):
        a0 = starts[i]
        fa = o3.gradient_force[a0:a0 + sizes[k]]
        fb = f1[starts[k]:starts[k] + sizes[k]]
        if i == k:
            assert torch.equal(fa, fb) and torch.equal(o3.energy[i], e1[k])
            same += 1
        else:
            assert (fa - fb).abs().max().item() <= 2e-5 * fscale
            moved += 1
    assert same >= 500 and moved >= 400, (same, moved)


@pytest.mark.parametrize('mode,whole', [(2, 1), (3, 1)])
def test_a_mixed_mode_below_the_persistent_regime_runs_the_whole_fused_form(mode, whole):
    """The one-direction modes exchange silu'(h) with the row path in the fragment order its PERSISTENT edge-MLP kernels keep; a small
    batch runs the row-local kernels (H row-major), so there the library runs both directions fused (pipeline.hip).  48 conformers:
    bitwise the result of the whole-form mode, three launches per direction, and next to the row path."""
    gen = torch.Generator().manual_seed(3)
    a = util.load_npz('aspirin_frames.npz')
    B, n = 48, 21
    pos = (torch.from_numpy(a['test0_pos']).float().repeat(B, 1) + 0.05 * torch.randn(B * n, 3, generator=gen)).cuda()
    z = torch.from_numpy(a['z']).long().repeat(B).cuda()
    batch = torch.repeat_interleave(torch.arange(B), n).cuda()
    cell = torch.zeros(B, 3, 3, device='cuda')
    model, _ = make_model('rand')
    with fused_mode(0):
        row = model(z, pos, cell, batch)
        f_row = row.gradient_force.clone()
    with fused_mode(whole):
        ref_out = model(z, pos, cell, batch)
        e_ref, f_ref = ref_out.energy.clone(), ref_out.gradient_force.clone()
    with fused_mode(mode):
        out, n_fwd, n_bwd = fused_launches(lambda: model(z, pos, cell, batch))
    assert (n_fwd, n_bwd) == (3, 3), (n_fwd, n_bwd)
    assert torch.equal(out.energy, e_ref) and torch.equal(out.gradient_force, f_ref)
    assert (out.gradient_force - f_row).abs().max().item() <= 2e-5


def test_fused_directions_swap_with_the_row_path_at_full_size():
    """BASELINE configs[1] at full size (1024 aspirin conformers; the row path runs its persistent edge-MLP kernels there, whose
    silu'(h) scratch the fused kernels share): fused both ways, forward only, adjoint only and not at all -- four results that
    must agree with each other to rounding and, on every 16th conformer, with the fp64 oracle."""
    from oracle import newtonnet_ref as ref
    a = util.load_npz('aspirin_frames.npz')
    B, n = 1024, 21
    gen = torch.Generator().manual_seed(0)
    pos = torch.from_numpy(a['test0_pos']).float().repeat(B, 1) + 0.05 * torch.randn(B * n, 3, generator=gen)
    z = torch.from_numpy(a['z']).long().repeat(B)
    batch = torch.repeat_interleave(torch.arange(B), n)
    model, sd = make_model('ckpt')
    args = (z.cuda(), pos.cuda(), torch.zeros(B, 3, 3, device='cuda'), batch.cuda())
    res = {}
    for mode, want in ((0, (0, 0)), (1, (3, 3)), (2, (3, 0)), (3, (0, 3))):
        with fused_mode(mode):
            out, n_fwd, n_bwd = fused_launches(lambda: model(*args))
            assert (n_fwd, n_bwd) == want, (mode, n_fwd, n_bwd)
            res[mode] = (out.energy.cpu().double(), out.gradient_force.cpu().double())
    pick = torch.arange(0, B, 16)
    ps = pos.view(B, n, 3)[pick].reshape(-1, 3).double()
    o = ref.energy_forces({k: v.double() for k, v in sd.items()}, z[:len(pick) * n], ps,
                          torch.zeros(len(pick), 3, 3, dtype=torch.float64), torch.repeat_interleave(torch.arange(len(pick)), n))
    e_ref = o['energy'].numpy()
    for mode, (e, f) in res.items():
        d = (f.view(B, n, 3)[pick].reshape(-1, 3) - o['forces']).abs()
        # next to the row path, conformer by conformer.  (A handful of the 1024 noisy conformers sit where the TRAINED model is badly
        # conditioned -- conformer 631 of this seed, max |F| 105 eV/A: the REFERENCE ITSELF run in fp32 differs from its own fp64 run by
        # max |dF| = 1.35e-3 eV/A there (measured with the reference imported in the build container; its neighbours sit
        # at 4-9e-6) -- so the bound on every conformer is statistical, the hard one is on the oracle-checked sample.)
        per = (f - res[0][1]).abs().view(B, -1).amax(dim=1)
        print(f'mode {mode}: force MAE vs fp64 {d.mean():.2e} max {d.max():.2e}; |dF| vs the row path: sample max {per[pick].max():.2e}, '
              f'all conformers mean {per.mean():.2e}, 99th percentile {per.quantile(0.99):.2e}, max {per.max():.2e}')
        assert d.mean() <= util.FORCE_MAE_TOL and d.max() <= util.FORCE_MAX_TOL
        assert np.all(np.abs(e[pick].numpy() - e_ref) <= util.energy_tol(e_ref))
        assert per[pick].max() <= 2e-5 and per.mean() <= 3e-6 and per.quantile(0.99) <= 1e-5
