"""Data-ingest mirror: extxyz reader and per-element statistics against the reference's own outputs
(tests/golden/case_statistics.npz was produced by running loader.py:MolecularStatistics)."""
import numpy as np
import torch

from newtonnet_amd.data import MolecularStatistics, collate, read_extxyz
from newtonnet_amd.layers import get_scaler_by_string, set_scaler_by_string
from tests import util


def write_xyz(path, z, frames_pos, energies, forces, lattice=None):
    sym = {1: 'H', 6: 'C', 8: 'O'}
    with open(path, 'w') as f:
        for p, e, fr in zip(frames_pos, energies, forces):
            f.write(f'{len(z)}\n')
            lat = f'Lattice="{" ".join(str(v) for v in lattice.reshape(-1))}" pbc="T T T" ' if lattice is not None else ''
            f.write(f'{lat}Properties=species:S:1:pos:R:3:forces:R:3 energy={float(e)!r}' + ('' if lattice is not None else ' pbc="F F F"') + '\n')
            for zi, pi, fi in zip(z, p, fr):
                f.write(f'{sym[int(zi)]} ' + ' '.join(repr(float(v)) for v in pi) + ' ' + ' '.join(repr(float(v)) for v in fi) + '\n')


def test_extxyz_roundtrip_matches_reference_data(tmp_path):
    a = util.load_npz('aspirin_frames.npz')          # values parsed from the reference's xyz by the golden generator
    path = tmp_path / 'a.xyz'
    write_xyz(path, a['z'], a['train_pos'], a['train_energy'], a['train_forces'])
    frames = read_extxyz(str(path))
    assert len(frames) == 8 and np.array_equal(frames[0].z, a['z'])
    z, pos, cell, batch, energy, force = collate(frames, torch.float64)
    assert np.array_equal(pos.numpy().reshape(8, 21, 3), a['train_pos'])
    assert np.array_equal(energy.numpy(), a['train_energy']) and np.array_equal(force.numpy().reshape(8, 21, 3), a['train_forces'])
    assert torch.all(cell == 0) and batch.tolist() == sum(([b] * 21 for b in range(8)), [])


def test_periodic_frame_is_wrapped_and_units(tmp_path):
    z = np.array([8, 1, 1])
    pos = np.array([[[-1.0, 12.5, 3.0], [0.5, 0.5, 0.5], [9.9, -0.2, 11.0]]])
    path = tmp_path / 'p.xyz'
    write_xyz(path, z, pos, [1.0], np.zeros((1, 3, 3)), lattice=np.diag([10.0, 10.0, 10.0]))
    fr = read_extxyz(str(path))[0]
    assert np.allclose(fr.pos, [[9.0, 2.5, 3.0], [0.5, 0.5, 0.5], [9.9, 9.8, 1.0]]) and np.allclose(fr.cell, np.diag([10.0] * 3))
    fr_b = read_extxyz(str(path), length_unit='Bohr', energy_unit='Ha')[0]
    assert np.allclose(fr_b.pos, fr.pos * 0.52917721067) and abs(fr_b.energy - 27.211386024367243) < 1e-12


def test_molecular_statistics_matches_reference():
    c = util.load_npz('case_statistics.npz')
    stats = MolecularStatistics()(torch.from_numpy(c['z']), torch.from_numpy(c['batch']), torch.from_numpy(c['energy']),
                                  torch.from_numpy(c['force']))
    np.testing.assert_allclose(stats['energy']['shift'].numpy(), c['energy_shift'], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(stats['energy']['scale'].numpy(), c['energy_scale'], rtol=1e-7)
    np.testing.assert_allclose(stats['force']['scale'].numpy(), c['force_scale'], rtol=1e-12)
    # feeds the scalers like newtonnet_train.py:88-90
    sc = set_scaler_by_string('energy', get_scaler_by_string('energy'), stats)
    assert sc.shift.weight.shape == (119, 1) and abs(sc.shift.weight[6, 0].item() - c['energy_shift'][6]) < 1e-3
