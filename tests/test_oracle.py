"""Pin the CPU oracle (oracle/newtonnet_ref.py) against the reference's own outputs.

The golden files were produced by running the reference in the build container
(tests/golden/gen_golden.py).  K1 / K2 are the only known answers the reference
ships."""
import numpy as np
import pytest
import torch

from oracle import newtonnet_ref as ref
from tests import util

CASES = ['aspirin1_rand', 'aspirin1_ckpt', 'aspirin8_rand', 'aspirin8_ckpt', 'ethanol4_rand', 'mixed_rand',
         'pbc216_rand', 'pbc_batch2_rand']


@pytest.mark.parametrize('case', CASES)
def test_oracle_matches_reference_fp64(case):
    z, pos, cell, batch, c = util.case_inputs(case)
    sd = util.load_state(case.split('_')[-1])
    out = ref.energy_forces(sd, z, pos, cell, batch, keep_intermediates=True)
    assert np.array_equal(out['edge_index'].numpy(), c['f64_edge_index'])       # bit-exact neighbor indices
    np.testing.assert_allclose(out['energy'].numpy(), c['f64_energy'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out['forces'].numpy(), c['f64_forces'], rtol=1e-10, atol=1e-12)
    if 'f64_dist_edge' in c:
        np.testing.assert_allclose(out['dist_edge'].numpy(), c['f64_dist_edge'], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(out['dir_edge'].numpy(), c['f64_dir_edge'], rtol=1e-12, atol=1e-14)
        for l, (a, f) in enumerate(out['layers']):
            np.testing.assert_allclose(a.numpy(), c[f'f64_atom_node_{l}'], rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(f.numpy(), c[f'f64_force_node_{l}'], rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize('case', CASES)
def test_oracle_matches_reference_fp32(case):
    z, pos, cell, batch, c = util.case_inputs(case, torch.float32)
    sd = util.load_state(case.split('_')[-1], torch.float32)
    out = ref.energy_forces(sd, z, pos, cell, batch)
    assert np.array_equal(out['edge_index'].numpy(), c['f32_edge_index'])
    assert np.all(np.abs(out['energy'].numpy().astype(np.float64) - c['f32_energy']) <= util.energy_tol(c['f32_energy']))
    # fp32-vs-fp32: same op sequence, threaded reductions may reorder -> small tolerance
    assert np.abs(out['forces'].numpy() - c['f32_forces']).max() < 2e-5


def test_zero_edge_molecules_do_not_crash():
    z, pos, cell, batch, c = util.case_inputs('mixed_rand')
    out = ref.energy_forces(util.load_state('rand'), z, pos, cell, batch)
    assert out['energy'].shape == (4,)
    ei = out['edge_index'].numpy()
    assert not np.isin(ei, [30, 31, 32]).any()          # lone atom and the 9 A pair have no edges
    assert np.all(out['forces'].numpy()[30:] == 0)


def test_K1_md_traj_fp32():
    """K1: scripts/md17_md/md.traj -- 201 frames written by the authors' CUDA fp32 run."""
    k = util.load_npz('kat_md_traj.npz')
    sd = util.load_state('ckpt', torch.float32)
    n_frames, n_atoms = k['positions'].shape[:2]
    z = torch.from_numpy(k['numbers']).long().repeat(n_frames)
    pos = torch.from_numpy(k['positions']).reshape(-1, 3).float()
    batch = torch.repeat_interleave(torch.arange(n_frames), n_atoms)
    out = ref.energy_forces(sd, z, pos, torch.zeros(n_frames, 3, 3), batch)
    e = out['energy'].numpy().astype(np.float64)
    assert np.all(np.abs(e - k['energy']) <= util.energy_tol(k['energy']))
    assert abs(e[0] - (-17591.826171875)) <= 2e-3       # md.log line 2 / K3
    df = np.abs(out['forces'].numpy().reshape(n_frames, n_atoms, 3) - k['forces'])
    assert df.max() < 5e-5 and df.mean() < 1e-5


def test_K2_logcsv_final_row_fp64():
    """K2: log.csv `final` row -- fp64 test-set MAEs of best_model.pt over the 500 test frames."""
    k = util.load_npz('kat_test_set.npz')
    sd = util.load_state('ckpt', torch.float64)
    n_frames, n_atoms = k['positions'].shape[:2]
    z = torch.from_numpy(k['z']).long().repeat(n_frames)
    pos = torch.from_numpy(k['positions']).reshape(-1, 3)
    batch = torch.repeat_interleave(torch.arange(n_frames), n_atoms)
    out = ref.energy_forces(sd, z, pos, torch.zeros(n_frames, 3, 3, dtype=torch.float64), batch)
    e_mae = np.abs(out['energy'].numpy() - k['energy']).mean()
    f_mae = np.abs(out['forces'].numpy().reshape(n_frames, n_atoms, 3) - k['forces']).mean()
    assert abs(e_mae - float(k['log_test_energy_mae'])) < 1e-11
    assert abs(f_mae - float(k['log_test_force_mae'])) < 1e-12
    assert out['edge_index'].shape[1] == 151366


def test_candidate_pair_order_unsorted_batch():
    batch = torch.tensor([1, 0, 1, 0, 2])
    p = ref.candidate_pairs(batch).numpy()
    assert p.T.tolist() == [[1, 3], [3, 1], [0, 2], [2, 0]]


def test_direct_force_head_pinned():
    """Oracle's direct_force head against the reference's own output for the same weights (gen_golden.py extra)."""
    c = util.load_npz('case_direct_force.npz')
    sd = {k[3:]: torch.from_numpy(v).double() for k, v in c.items() if k.startswith('sd.')}
    z, pos, cell, batch = (torch.from_numpy(c[k]) for k in ('z', 'pos', 'cell', 'batch'))
    out = ref.energy_forces(sd, z, pos.double(), cell.double(), batch)
    df = ref.direct_force_head(sd, 2, out['atom_node'], out['force_node'], z)
    np.testing.assert_allclose(df.numpy(), c['direct_force'], rtol=1e-4, atol=1e-6)      # reference ran in fp32
    np.testing.assert_allclose(out['forces'].numpy(), c['forces'], atol=5e-5)


def test_layer_norm_pinned():
    """layer_norm=True (newtonnet.py:202-205,228-231): the oracle against the reference's own fp64 output for the fp32 weights
    stored in the fixture (gen_golden.py layernorm)."""
    c = util.load_npz('case_layernorm.npz')
    sd = {k[3:]: torch.from_numpy(v).double() for k, v in c.items() if k.startswith('sd.')}
    assert 'interaction_layers.0.layer_norm.weight' in sd
    z, pos, cell, batch = (torch.from_numpy(c[k]) for k in ('z', 'pos', 'cell', 'batch'))
    out = ref.energy_forces(sd, z, pos.double(), cell.double(), batch)
    np.testing.assert_allclose(out['energy'].numpy(), c['energy'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out['forces'].numpy(), c['forces'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out['atom_node'].numpy(), c['atom_node'], rtol=0, atol=1e-12)


VIRIAL_CASES = ['aspirin8', 'pbc216', 'pbc_batch2', 'triclinic64']


@pytest.mark.parametrize('case', VIRIAL_CASES)
def test_oracle_virial_stress_pinned(case):
    """virial / stress heads (output.py:154-180, strain construction newtonnet.py:146-155): the oracle's autograd virial
    against the reference's own output for ['energy','gradient_force','virial','stress'] (gen_golden.py virial)."""
    c = util.load_npz(f'case_virial_{case}.npz')
    sd = util.load_state('rand')
    z, batch = torch.from_numpy(c['z']).long(), torch.from_numpy(c['batch']).long()
    pos, cell = torch.from_numpy(c['pos']), torch.from_numpy(c['cell'])
    out = ref.energy_forces(sd, z, pos, cell, batch)
    assert np.array_equal(out['edge_index'].numpy(), c['f64_edge_index'])
    np.testing.assert_allclose(out['energy'].numpy(), c['f64_energy'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out['forces'].numpy(), c['f64_forces'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out['virial'].numpy(), c['f64_virial'], rtol=1e-10, atol=1e-11)
    if 'f64_stress' in c:      # stress = dE/d(strain) / det(cell) = -virial / volume (output.py:175-179)
        stress = -out['virial'] / cell.det().view(-1, 1, 1)
        np.testing.assert_allclose(stress.numpy(), c['f64_stress'], rtol=1e-10, atol=1e-13)
    # fp32 run of the reference vs the fp32 oracle
    out32 = ref.energy_forces(ref.cast_state(sd, torch.float32), z, pos.float(), cell.float(), batch)
    assert np.array_equal(out32['edge_index'].numpy(), c['f32_edge_index'])
    scale = max(1.0, float(np.abs(c['f64_virial']).max()))
    assert np.abs(out32['virial'].numpy() - c['f32_virial']).max() < 5e-5 * scale


def test_oracle_boundary_cases_fp32():
    """Neighbor predicate at the boundary (representations.py:85-98, fp32): pairs at r (1 +- k ulp) and periodic pairs at a
    fractional separation of +-0.5 +- k ulp -- the oracle's fp32 radius_graph against the reference's own edge_index."""
    c = util.load_npz('case_boundary.npz')
    r = float(c['cutoff'])
    n_a = int(c['a_batch'].max()) + 1
    ei, d = ref.radius_graph(torch.from_numpy(c['a_pos']), torch.zeros(n_a, 3, 3), torch.from_numpy(c['a_batch']), r)
    assert np.array_equal(ei.numpy(), c['a_edge_index']) and np.array_equal(d.numpy(), c['a_disp'])
    ei, d = ref.radius_graph(torch.from_numpy(c['b_pos']), torch.from_numpy(c['b_cell']), torch.from_numpy(c['b_batch']), r)
    assert np.array_equal(ei.numpy(), c['b_edge_index']) and np.array_equal(d.numpy(), c['b_disp'])
    # the fixture is adversarial: almost every open-boundary pair is within 4 ulp of the cutoff, on both sides of it
    pa = torch.from_numpy(c['a_pos'])
    dn = (pa[0::2] - pa[1::2]).norm(dim=1)
    ties = int(((dn - r).abs() <= 4 * np.spacing(np.float32(r))).sum())
    inside = c['a_edge_index'].shape[1] // 2
    assert ties > 1500 and 0.2 < inside / (len(pa) // 2) < 0.8


@pytest.mark.parametrize('tag,name,p', [('cosine', 'cosine', 9), ('poly6', 'polynomial', 6)])
def test_oracle_other_envelopes_pinned(tag, name, p):
    """CosineCutoff (representations.py:177-203) and PolynomialCutoff(p != 9) swapped into the edge embedding: the oracle against
    the reference's own output (gen_golden.py cosine)."""
    c = util.load_npz('case_envelope.npz')
    sd = util.load_state('rand')
    z, batch = torch.from_numpy(c['z']).long(), torch.from_numpy(c['batch']).long()
    ref.set_envelope(name, p)
    try:
        out = ref.energy_forces(sd, z, torch.from_numpy(c['pos']), torch.from_numpy(c['cell']), batch)
    finally:
        ref.set_envelope()
    assert np.array_equal(out['edge_index'].numpy(), c['edge_index'])
    np.testing.assert_allclose(out['dist_edge'].numpy(), c[f'{tag}_dist_edge'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(out['energy'].numpy(), c[f'{tag}_energy'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out['forces'].numpy(), c[f'{tag}_forces'], rtol=1e-10, atol=1e-12)


def test_oracle_training_gradients_pinned_to_reference():
    """The training objective and every parameter gradient of the REFERENCE (its own model in train mode, its own loss factory
    with the published weights, loss.backward(): trainer.py:299-313; generated by tests/golden/gen_golden.py train) against the
    oracle's restatement -- pins the checker of the hand-written training path (row T)."""
    c = util.load_npz('case_train_mixed.npz')
    z, pos, cell, batch, _ = util.case_inputs('mixed_rand', torch.float64)
    sd = util.load_state('rand', torch.float64)
    loss, grads = ref.training_loss_grads(sd, z, pos, cell, batch, torch.from_numpy(c['energy_label']).double(),
                                          torch.from_numpy(c['force_label']).double())
    assert abs(loss.item() - float(c['loss'])) <= 1e-10 * abs(float(c['loss']))
    names = [k[5:] for k in c if k.startswith('grad.')]
    assert len(names) == 39
    for name in names:
        want, nrm = c['grad.' + name].astype(np.float64), float(c['gnorm.' + name])
        got = grads[name].numpy()
        assert abs(np.linalg.norm(got) - nrm) <= 1e-9 * max(nrm, 1e-30), name                      # the float64 norm
        assert np.abs(got - want).max() <= 2e-7 * max(np.abs(want).max(), 1e-30), name              # float32-stored entries


def _fuzz_bits(edge_index, n):
    bits = np.zeros((n, 2), dtype=np.uint8)
    ei = edge_index.cpu().numpy()
    bits[ei[0] // 2, ei[0] % 2] = 1
    return bits


def test_oracle_triclinic_fuzz_fp32():
    """General triclinic cells at the decision boundaries (tests/util.py:triclinic_fuzz_inputs; expected bits from the
    reference's own RadiusGraph, gen_golden.py triclinic_fuzz): the inputs regenerate bit for bit from the seed and the oracle's
    radius graph -- the same torch CPU ops -- agrees with the reference on every one of the first 20 000 cells."""
    import hashlib
    c = util.load_npz('case_triclinic_fuzz.npz')
    n = int(c['n'])
    pos, cells, batch, kinds = util.triclinic_fuzz_inputs(n, int(c['seed']), float(c['cutoff']))
    assert hashlib.sha256(pos.tobytes() + cells.tobytes()).hexdigest() == str(c['input_sha256'])
    want = np.unpackbits(c['bits'])[:2 * n].reshape(n, 2)
    m = 20000
    ei, _ = ref.radius_graph(torch.from_numpy(pos[:2 * m]), torch.from_numpy(cells[:m]), torch.from_numpy(batch[:2 * m]),
                             float(c['cutoff']))
    assert np.array_equal(_fuzz_bits(ei, m), want[:m])


def test_oracle_direct_force_training_pinned_to_reference():
    """['energy', 'direct_force'] training (no derivative head: trainer.py:299-313, DirectForceLoss loss.py:41-47) -- the
    reference's own loss factory {'energy': mse, 'direct_force': mse x 20} and loss.backward() (gen_golden.py train_direct)
    against the oracle's restatement: loss, predictions and all 46 parameter gradients."""
    sd, c = util.direct_train_state(torch.float64)
    z, pos, cell, batch, _ = util.case_inputs('mixed_rand', torch.float64)
    loss, grads = ref.training_loss_grads(sd, z, pos, cell, batch, torch.from_numpy(c['energy_label']).double(), None,
                                          direct_head=1, direct_label=torch.from_numpy(c['force_label']).double(), w_direct=20.0)
    assert abs(loss.item() - float(c['loss'])) <= 1e-10 * abs(float(c['loss']))
    names = [k[5:] for k in c if k.startswith('grad.')]
    assert len(names) == 46
    for name in names:
        want = c['grad.' + name].astype(np.float64)
        assert np.abs(grads[name].numpy() - want).max() <= 2e-7 * max(np.abs(want).max(), 1e-30), name
