#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, read-only).  The
reference's Python never enters this repo: this script imports it in place,
feeds it inputs, and stores inputs + outputs as .npz data.

Two third-party modules the reference imports are absent from this image and
are replaced by minimal stand-ins *for the import only* (SURVEY.md 8c):
  * torch_geometric.utils.scatter  -> sum-scatter via index_add_ (the only
    reduce the reference ever uses on this path; newtonnet.py:214,226, output.py:246)
  * les.Les                        -> empty nn.Module (only constructed, never
    called, unless a 'charge' head exists; output.py:227-231)

Fixtures written:
  aspirin_frames.npz     z, train frames 0-7 + test frame 0 positions/energies/forces (data from the xyz files)
  ckpt_state.npz         the shipped best_model.pt parameters, renamed to the current key layout, fp32+fp64 views
  case_*.npz             inputs + reference outputs (edge_index, dist_edge, dir_edge, per-layer nodes, energy, force)
  kat_md_traj.npz        K1: 201 frames of md.traj (positions, energy, forces)
  kat_test_set.npz       K2: 500 test frames (positions, energies, forces) + the log.csv MAEs
"""
import io
import json
import math
import os
import pickle
import struct
import sys
import types

import numpy as np
import torch
from torch import nn

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


# ----------------------------------------------------------------------------
# import the reference with the two stand-ins
# ----------------------------------------------------------------------------
def _install_shims():
    def scatter(src, index, dim=0, dim_size=None, reduce='sum'):
        assert reduce in ('sum', 'mean') and dim == 0     # 'mean' only in the data statistics (loader.py:224)
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        out = src.new_zeros((dim_size,) + tuple(src.shape[1:])).index_add_(0, index, src)
        if reduce == 'mean':
            cnt = src.new_zeros(dim_size).index_add_(0, index, torch.ones_like(index, dtype=src.dtype)).clamp(min=1)
            out = out / cnt.reshape((-1,) + (1,) * (src.dim() - 1))
        return out

    tg = types.ModuleType('torch_geometric')
    tgu = types.ModuleType('torch_geometric.utils')
    tgu.scatter = scatter
    tg.utils = tgu
    sys.modules['torch_geometric'] = tg
    sys.modules['torch_geometric.utils'] = tgu

    class Les(nn.Module):
        def __init__(self):
            super().__init__()
            self.atomwise = nn.Identity()
            self.ewald = nn.Identity()
            self.bec = nn.Identity()

    les = types.ModuleType('les')
    les.Les = Les
    sys.modules['les'] = les


def import_reference():
    _install_shims()
    sys.path.insert(0, REF)
    from newtonnet.models.newtonnet import NewtonNet  # noqa
    return NewtonNet


# ----------------------------------------------------------------------------
# data readers (extxyz, ASE ULM .traj) -- written here, no ase in this image
# ----------------------------------------------------------------------------
SYM2Z = {'H': 1, 'C': 6, 'N': 7, 'O': 8}


def read_extxyz(path, max_frames=None):
    frames = []
    with open(path) as f:
        while True:
            line = f.readline()
            if not line.strip():
                break
            n = int(line)
            header = f.readline()
            energy = float(header.split('energy=')[1].split()[0])
            z, pos, frc = [], [], []
            for _ in range(n):
                t = f.readline().split()
                z.append(SYM2Z[t[0]])
                pos.append([float(v) for v in t[1:4]])
                frc.append([float(v) for v in t[4:7]])
            frames.append((np.array(z), np.array(pos), energy, np.array(frc)))
            if max_frames and len(frames) >= max_frames:
                break
    return frames


def read_ulm(path):
    """Minimal reader for ASE's ULM container ('- of Ulm' magic)."""
    raw = open(path, 'rb').read()
    assert raw[:8] == b'- of Ulm'
    version, nitems, pos0 = struct.unpack('<qqq', raw[24:48])
    offsets = np.frombuffer(raw, '<i8', nitems, pos0)

    def resolve(obj, base):
        if isinstance(obj, dict):
            if 'ndarray' in obj and len(obj) == 1:
                shape, dtype, off = obj['ndarray']
                cnt = int(np.prod(shape)) if shape else 1
                return np.frombuffer(raw, np.dtype(dtype), cnt, off).reshape(shape).copy()
            return {(k[:-1] if k.endswith('.') else k): resolve(v, base) for k, v in obj.items()}
        return obj

    items = []
    for off in offsets:
        (ln,) = struct.unpack('<q', raw[off:off + 8])
        items.append(resolve(json.loads(raw[off + 8:off + 8 + ln].decode()), off))
    return items


# ----------------------------------------------------------------------------
# checkpoint: old-layout whole-module pickle -> current key names
# ----------------------------------------------------------------------------
class _StubUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith('newtonnet'):
            return type(name, (nn.Module,), {'__module__': module})
        return super().find_class(module, name)


class _StubPickle:
    Unpickler = _StubUnpickler
    load = staticmethod(lambda f, **kw: _StubUnpickler(f, **kw).load())
    __name__ = 'stub_pickle'


def load_checkpoint_state():
    path = f'{REF}/scripts/md17_model/training_1/models/best_model.pt'
    mod = torch.load(path, map_location='cpu', weights_only=False, pickle_module=_StubPickle)
    sd = {}
    for k, v in mod.state_dict().items():
        k = k.replace('embedding_layer.edge_embedding.frequencies',
                      'embedding_layers.edge_embedding.embedding.frequencies')
        k = k.replace('embedding_layer.', 'embedding_layers.')
        sd[k] = v.detach().clone()
    return sd


# ----------------------------------------------------------------------------
# run the reference
# ----------------------------------------------------------------------------
def run_reference(NewtonNet, sd, z, pos, cell, batch, dtype, n_features=128, n_basis=20, n_interactions=3,
                  cutoff=5.0):
    model = NewtonNet(cutoff=cutoff, n_features=n_features, n_basis=n_basis, n_interactions=n_interactions,
                      output_properties=['energy', 'gradient_force'])
    model.to(torch.float64)      # widen first: load_state_dict copies INTO the existing (fp32) parameters
    model.load_state_dict({k: v.to(torch.float64) for k, v in sd.items()}, strict=True)
    model.to(dtype)
    model.eval()
    pos = pos.to(dtype).clone()
    cell = cell.to(dtype).clone()
    layers = []
    hooks = [il.register_forward_hook(lambda m, i, o: layers.append((o[0].detach().clone(), o[1].detach().clone())))
             for il in model.interaction_layers]
    edge = {}
    h2 = model.embedding_layers.edge_embedding.register_forward_hook(
        lambda m, i, o: edge.update(dist_edge=o[0].detach().clone(), dir_edge=o[1].detach().clone()))
    out = model(z, pos, cell, batch)
    for h in hooks + [h2]:
        h.remove()
    res = dict(energy=out.energy.detach(), forces=out.gradient_force.detach(), edge_index=out.edge_index,
               dist_edge=edge['dist_edge'], dir_edge=edge['dir_edge'])
    for l, (a, f) in enumerate(layers):
        res[f'atom_node_{l}'] = a
        res[f'force_node_{l}'] = f
    return res


def to_np(d):
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def random_state_via_reference(NewtonNet, seed, **kw):
    """Seeded random weights exactly as the reference constructs them (torch.manual_seed + default init)."""
    torch.manual_seed(seed)
    model = NewtonNet(output_properties=['energy', 'gradient_force'], **kw)
    model.to(torch.float64)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def periodic_box(n_side, spacing, jitter, seed, species=(1, 6, 7, 8)):
    """SURVEY 8d config-5 recipe at a small size: simple-cubic lattice + uniform jitter."""
    g = torch.Generator().manual_seed(seed)
    idx = torch.arange(n_side ** 3)
    grid = torch.stack([idx // (n_side * n_side), (idx // n_side) % n_side, idx % n_side], 1).double() * spacing
    pos = grid + (torch.rand(grid.shape, generator=g, dtype=torch.float64) - 0.5) * 2 * jitter
    z = torch.tensor(species)[torch.randint(0, len(species), (n_side ** 3,), generator=g)]
    L = n_side * spacing
    pos = pos % L
    return z, pos, torch.eye(3, dtype=torch.float64).unsqueeze(0) * L


FULL_CASES = ('aspirin1', 'ethanol4', 'mixed')


def main():
    NewtonNet = import_reference()
    os.makedirs(OUT, exist_ok=True)

    # -- raw data -------------------------------------------------------------
    train = read_extxyz(f'{REF}/scripts/md17_data/aspirin/ccsd_train/raw/aspirin_ccsd-train.xyz', 8)
    test = read_extxyz(f'{REF}/scripts/md17_data/aspirin/ccsd_test/raw/aspirin_ccsd-test.xyz')
    z_asp = train[0][0]
    np.savez_compressed(f'{OUT}/aspirin_frames.npz', z=z_asp,
                        train_pos=np.stack([f[1] for f in train]), train_energy=np.array([f[2] for f in train]),
                        train_forces=np.stack([f[3] for f in train]),
                        test0_pos=test[0][1], test0_energy=test[0][2], test0_forces=test[0][3])

    # -- checkpoint -----------------------------------------------------------
    ck = load_checkpoint_state()
    np.savez_compressed(f'{OUT}/ckpt_state.npz', **{k: v.to(torch.float64).numpy() for k, v in ck.items()})
    print('checkpoint params:', sum(v.numel() for v in ck.values()))

    rnd = random_state_via_reference(NewtonNet, 0, cutoff=5.0, n_features=128, n_basis=20, n_interactions=3)
    # constructed in fp32 then widened, so fp32 storage is lossless
    assert all(torch.equal(v, v.float().double()) for v in rnd.values())
    np.savez_compressed(f'{OUT}/rand_state_seed0.npz', **{k: v.float().numpy() for k, v in rnd.items()})

    zt = torch.tensor(z_asp, dtype=torch.long)

    def mol_batch(frames_pos):
        B = len(frames_pos)
        z = zt.repeat(B)
        pos = torch.tensor(np.concatenate(frames_pos), dtype=torch.float64)
        batch = torch.repeat_interleave(torch.arange(B), len(z_asp))
        return z, pos, torch.zeros(B, 3, 3, dtype=torch.float64), batch

    cases = {}
    # 1 aspirin frame / 8-frame batch, random + checkpoint weights
    cases['aspirin1'] = mol_batch([test[0][1]])
    cases['aspirin8'] = mol_batch([f[1] for f in train])
    # ethanol-shaped 9-atom molecules, batch of 4
    eth0 = torch.tensor([[0.00, 0.00, 0.00], [1.52, 0.00, 0.00], [2.05, 1.32, 0.00],
                         [-0.39, 1.02, 0.00], [-0.39, -0.51, 0.89], [-0.39, -0.51, -0.89],
                         [1.90, -0.53, 0.88], [1.90, -0.53, -0.88], [3.01, 1.30, 0.00]], dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    eth = [(eth0 + 0.1 * torch.randn(9, 3, generator=g, dtype=torch.float64)).numpy() for _ in range(4)]
    cases['ethanol4'] = (torch.tensor([6, 6, 8, 1, 1, 1, 1, 1, 1]).repeat(4),
                         torch.tensor(np.concatenate(eth)), torch.zeros(4, 3, 3, dtype=torch.float64),
                         torch.repeat_interleave(torch.arange(4), 9))
    # mixed sizes: aspirin (21) + ethanol (9) + a single atom (zero-edge molecule) + a far-apart pair (no edges)
    far = np.array([[0.0, 0, 0], [9.0, 0, 0]])
    cases['mixed'] = (torch.cat([zt, torch.tensor([6, 6, 8, 1, 1, 1, 1, 1, 1]), torch.tensor([8]), torch.tensor([1, 1])]),
                      torch.tensor(np.concatenate([train[1][1], eth[0], np.zeros((1, 3)), far])),
                      torch.zeros(4, 3, 3, dtype=torch.float64),
                      torch.tensor([0] * 21 + [1] * 9 + [2] + [3, 3]))
    # periodic orthorhombic box 6^3 = 216 atoms, L = 12.77 (> 2r): reference PBC == true minimum image
    zb, pb, cb = periodic_box(6, 100.0 / 47.0, 0.5, 0)
    cases['pbc216'] = (zb, pb, cb, torch.zeros(216, dtype=torch.long))
    # two periodic boxes of different (orthorhombic) shape in one batch
    zb2, pb2, _ = periodic_box(5, 2.4, 0.4, 1)
    cb2 = torch.diag(torch.tensor([12.0, 12.0, 12.0], dtype=torch.float64)).unsqueeze(0)
    cases['pbc_batch2'] = (torch.cat([zb, zb2]), torch.cat([pb, pb2]), torch.cat([cb, cb2]),
                           torch.cat([torch.zeros(216, dtype=torch.long), torch.ones(125, dtype=torch.long)]))

    for name, (z, pos, cell, batch) in cases.items():
        for wname, sd in (('rand', rnd), ('ckpt', ck)):
            if wname == 'ckpt' and not name.startswith('aspirin'):
                continue
            rec = dict(z=z.numpy(), pos=pos.numpy(), cell=cell.numpy(), batch=batch.numpy())
            for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
                r = run_reference(NewtonNet, sd, z, pos, cell, batch, dt)
                for k, v in to_np(r).items():
                    big = k.startswith(('atom_node', 'force_node', 'dist_edge', 'dir_edge'))
                    if big and (tag == 'f32' or name not in FULL_CASES):
                        continue  # per-layer / per-edge tensors: fp64 only, small cases only (fixture size)
                    rec[f'{tag}_{k}'] = v
            np.savez_compressed(f'{OUT}/case_{name}_{wname}.npz', **rec)
            print(name, wname, 'E =', rec['f64_energy'][:2], 'edges =', rec['f64_edge_index'].shape[1],
                  'f32 edges equal:', np.array_equal(rec['f64_edge_index'], rec['f32_edge_index']))

    # -- K1: md.traj ------------------------------------------------------------
    items = read_ulm(f'{REF}/scripts/md17_md/md.traj')
    numbers = items[0]['numbers']
    P = np.stack([it['positions'] for it in items])
    E = np.array([it['calculator']['energy'] for it in items])
    Fr = np.stack([it['calculator']['forces'] for it in items])
    np.savez_compressed(f'{OUT}/kat_md_traj.npz', numbers=numbers, positions=P, energy=E, forces=Fr)
    print('K1 frames', len(items), 'E0', E[0])

    # -- K2: test set + log.csv final row ------------------------------------------
    import csv
    rows = list(csv.DictReader(open(f'{REF}/scripts/md17_model/training_1/log.csv')))
    final = rows[-1]
    np.savez_compressed(f'{OUT}/kat_test_set.npz', z=test[0][0], positions=np.stack([f[1] for f in test]),
                        energy=np.array([f[2] for f in test]), forces=np.stack([f[3] for f in test]),
                        log_test_energy_mae=float(final['test_energy_mae']),
                        log_test_force_mae=float(final['test_gradient_force_mae']))
    print('K2', final['test_energy_mae'], final['test_gradient_force_mae'])


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


# ----------------------------------------------------------------------------
# "next" rows of SURVEY.md 8(f): direct_force head (output.py:115-132) and per-element statistics (loader.py:197-230)
# ----------------------------------------------------------------------------
def main_extra():
    NewtonNet = import_reference()
    train = read_extxyz(f'{REF}/scripts/md17_data/aspirin/ccsd_train/raw/aspirin_ccsd-train.xyz', 64)
    z_asp = torch.tensor(train[0][0], dtype=torch.long)

    # ---- direct_force head: seeded reference model with ['energy', 'gradient_force', 'direct_force']
    torch.manual_seed(1)
    model = NewtonNet(output_properties=['energy', 'gradient_force', 'direct_force'])
    model.to(torch.float64)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.eval()
    B = 4
    z = z_asp.repeat(B)
    pos = torch.tensor(np.concatenate([f[1] for f in train[:B]]), dtype=torch.float64)
    batch = torch.repeat_interleave(torch.arange(B), 21)
    cell = torch.zeros(B, 3, 3, dtype=torch.float64)
    out = model(z, pos.clone(), cell, batch)
    np.savez_compressed(f'{OUT}/case_direct_force.npz', z=z.numpy(), pos=pos.numpy(), cell=cell.numpy(), batch=batch.numpy(),
                        energy=out.energy.detach().numpy(), forces=out.gradient_force.detach().numpy(),
                        direct_force=out.direct_force.detach().numpy(),
                        **{'sd.' + k: v.float().numpy() for k, v in sd.items()})
    print('direct_force', out.direct_force.shape, float(out.direct_force.abs().max()))

    # ---- MolecularStatistics on the first 64 train frames (loader.py:197-230); the class needs only torch + scatter,
    # but its module imports ase / torch_geometric.data at import time -> provide empty stand-ins for the import
    for name in ('ase', 'ase.io', 'tqdm', 'torch_geometric.data'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['ase'].units = types.SimpleNamespace(kcal=1.0, mol=1.0, kJ=1.0, __setattr__=lambda *a: None)
    sys.modules['ase.io'].read = None
    sys.modules['tqdm'].tqdm = lambda x, **k: x
    for cls in ('Dataset', 'InMemoryDataset', 'Data'):
        setattr(sys.modules['torch_geometric.data'], cls, type(cls, (), {}))
    units = types.ModuleType('ase.units')
    units.kcal = units.mol = units.kJ = 1.0
    sys.modules['ase'].units = units
    sys.modules['ase.units'] = units
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_loader', f'{REF}/newtonnet/data/loader.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n = 64
    data = types.SimpleNamespace(
        z=z_asp.repeat(n), batch=torch.repeat_interleave(torch.arange(n), 21),
        energy=torch.tensor([f[2] for f in train[:n]], dtype=torch.float64),
        force=torch.tensor(np.concatenate([f[3] for f in train[:n]]), dtype=torch.float64))
    stats = mod.MolecularStatistics()(data)
    np.savez_compressed(f'{OUT}/case_statistics.npz', z=data.z.numpy(), batch=data.batch.numpy(), energy=data.energy.numpy(),
                        force=data.force.numpy(), energy_shift=stats['energy']['shift'].numpy(),
                        energy_scale=stats['energy']['scale'].numpy(), force_scale=stats['force']['scale'].numpy())
    print('stats shift', stats['energy']['shift'][[1, 6, 8]], 'scale', stats['energy']['scale'][[1, 6, 8]])


def main_layernorm():
    """layer_norm=True (newtonnet.py:202-205,228-231): seeded reference model with non-trivial LayerNorm affine parameters;
    weights are rounded to fp32 first so the fp64 reference run uses exactly the values stored in the fixture."""
    NewtonNet = import_reference()
    train = read_extxyz(f'{REF}/scripts/md17_data/aspirin/ccsd_train/raw/aspirin_ccsd-train.xyz', 8)
    torch.manual_seed(2)
    model = NewtonNet(layer_norm=True, output_properties=['energy', 'gradient_force'])
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if 'layer_norm.weight' in name:
                prm.add_(0.2 * torch.randn(prm.shape, generator=g))
            if 'layer_norm.bias' in name:
                prm.add_(0.1 * torch.randn(prm.shape, generator=g))
    model.to(torch.float32)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(torch.float64)
    model.eval()
    B = 4
    z = torch.tensor(train[0][0], dtype=torch.long).repeat(B)
    pos = torch.tensor(np.concatenate([f[1] for f in train[:B]]), dtype=torch.float64)
    batch = torch.repeat_interleave(torch.arange(B), 21)
    cell = torch.zeros(B, 3, 3, dtype=torch.float64)
    out = model(z, pos.clone(), cell, batch)
    np.savez_compressed(f'{OUT}/case_layernorm.npz', z=z.numpy(), pos=pos.numpy(), cell=cell.numpy(), batch=batch.numpy(),
                        energy=out.energy.detach().numpy(), forces=out.gradient_force.detach().numpy(),
                        atom_node=out.atom_node.detach().numpy(),
                        **{'sd.' + k: v.float().numpy() for k, v in sd.items()})
    print('layernorm', float(out.energy.abs().max()), float(out.gradient_force.abs().max()),
          [k for k in sd if 'layer_norm' in k][:2])


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'extra':
    main_extra()
if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'layernorm':
    main_layernorm()


# ----------------------------------------------------------------------------
# virial / stress heads (output.py:154-180; strain construction newtonnet.py:146-155)
# ----------------------------------------------------------------------------
def triclinic_box(seed=5):
    """64 atoms in a triclinic cell (rows = lattice vectors), every lattice vector and height > 2 r."""
    g = torch.Generator().manual_seed(seed)
    cell = torch.tensor([[11.5, 0.0, 0.0], [2.3, 11.0, 0.0], [-1.7, 2.9, 10.8]], dtype=torch.float64)
    idx = torch.arange(64)
    frac = (torch.stack([idx // 16, (idx // 4) % 4, idx % 4], 1).double() + 0.5) / 4.0
    frac = (frac + (torch.rand(64, 3, generator=g, dtype=torch.float64) - 0.5) * 0.12) % 1.0
    pos = frac @ cell
    z = torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (64,), generator=g)]
    return z, pos, cell.unsqueeze(0)


def main_virial():
    """Reference run with output_properties ['energy','gradient_force','virial','stress'] on aspirin8, pbc216, pbc_batch2 and a
    triclinic box -> case_virial_<name>.npz (fp64 and fp32 runs).  stress is stored for the periodic cases only (the
    reference divides by det(cell) = 0 otherwise)."""
    NewtonNet = import_reference()
    rnd = {k: torch.from_numpy(v).double() for k, v in np.load(f'{OUT}/rand_state_seed0.npz').items()}
    train = read_extxyz(f'{REF}/scripts/md17_data/aspirin/ccsd_train/raw/aspirin_ccsd-train.xyz', 8)
    z_asp = torch.tensor(train[0][0], dtype=torch.long)
    cases = {}
    cases['aspirin8'] = (z_asp.repeat(8), torch.tensor(np.concatenate([f[1] for f in train])),
                         torch.zeros(8, 3, 3, dtype=torch.float64), torch.repeat_interleave(torch.arange(8), 21))
    zb, pb, cb = periodic_box(6, 100.0 / 47.0, 0.5, 0)
    cases['pbc216'] = (zb, pb, cb, torch.zeros(216, dtype=torch.long))
    zb2, pb2, _ = periodic_box(5, 2.4, 0.4, 1)
    cb2 = torch.diag(torch.tensor([12.0, 12.0, 12.0], dtype=torch.float64)).unsqueeze(0)
    cases['pbc_batch2'] = (torch.cat([zb, zb2]), torch.cat([pb, pb2]), torch.cat([cb, cb2]),
                           torch.cat([torch.zeros(216, dtype=torch.long), torch.ones(125, dtype=torch.long)]))
    zt, pt, ct = triclinic_box()
    cases['triclinic64'] = (zt, pt, ct, torch.zeros(64, dtype=torch.long))
    props = ['energy', 'gradient_force', 'virial', 'stress']
    for name, (z, pos, cell, batch) in cases.items():
        rec = dict(z=z.numpy(), pos=pos.numpy(), cell=cell.numpy(), batch=batch.numpy())
        for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
            model = NewtonNet(output_properties=props)
            model.to(torch.float64)
            sd = {k: v for k, v in rnd.items()}
            missing = model.load_state_dict(sd, strict=False)   # virial / stress heads have no parameters
            assert not missing.unexpected_keys and all('scalers' in k or 'output_layers' in k for k in missing.missing_keys), missing
            model.to(dt)
            model.eval()
            out = model(z, pos.to(dt).clone(), cell.to(dt).clone(), batch)
            rec[f'{tag}_energy'] = out.energy.detach().numpy()
            rec[f'{tag}_forces'] = out.gradient_force.detach().numpy()
            rec[f'{tag}_virial'] = out.virial.detach().numpy()
            if bool((cell != 0).any()):
                rec[f'{tag}_stress'] = out.stress.detach().numpy()
            rec[f'{tag}_edge_index'] = out.edge_index.numpy()
        np.savez_compressed(f'{OUT}/case_virial_{name}.npz', **rec)
        print('virial', name, 'E', rec['f64_energy'][:2], '|virial|max', np.abs(rec['f64_virial']).max(),
              'edges', rec['f64_edge_index'].shape[1],
              'f32 == f64 edges:', np.array_equal(rec['f64_edge_index'], rec['f32_edge_index']))


# ----------------------------------------------------------------------------
# neighbor-predicate boundary cases (representations.py:85-98: strict `norm < r` in the model dtype; PBC round() at +-0.5)
# ----------------------------------------------------------------------------
def _ulp_steps(x, k):
    """x moved by k fp32 ulps (k may be negative)."""
    v = np.float32(x)
    for _ in range(abs(int(k))):
        v = np.nextafter(v, np.float32(np.inf if k > 0 else -np.inf))
    return v


def main_boundary():
    """Adversarial fp32 inputs for RadiusGraph: the reference's own fp32 edge_index for
      (a) 2-atom molecules whose separation is r (1 + k ulp), k in {-4..4}, along the axes, along (3,4,0)/5 and along
          random directions, with and without a random common offset (so the subtraction rounds);
      (b) periodic pairs whose fractional coordinate sits at +-0.5 +- k ulp (round-half-even), in orthorhombic and
          triclinic cells whose lattice vector has length exactly 2 r, so that the image choice and the `< r` test tie
          at the same time.
    The fixture stores the fp32 inputs and the reference's edge_index (fp32 run); the HIP neighbor list must reproduce it
    bit for bit (tests/test_hip_parity.py::test_neighbor_boundary_cases)."""
    NewtonNet = import_reference()
    from newtonnet.layers.representations import RadiusGraph
    r = 5.0
    rg = RadiusGraph(r)
    g = torch.Generator().manual_seed(11)
    # ---- (a) open boundary
    pos, batch = [], []
    b = 0

    def add_pair(p0, p1):
        nonlocal b
        pos.extend([p0, p1])
        batch.extend([b, b])
        b += 1

    dirs = [np.array(d, dtype=np.float64) for d in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [0.6, 0.8, 0], [0, 0.6, 0.8],
                                                      [0.8, 0, 0.6])]
    for _ in range(120):
        v = torch.randn(3, generator=g, dtype=torch.float64).numpy()
        dirs.append(v / np.linalg.norm(v))
    for d in dirs:
        for k in (-4, -2, -1, 0, 1, 2, 4):
            for off in (False, True):
                o = (torch.randn(3, generator=g, dtype=torch.float64).numpy() * 3.0) if off else np.zeros(3)
                p0 = o.astype(np.float32)
                target = np.float64(_ulp_steps(r, k))
                p1 = (p0.astype(np.float64) + d * target).astype(np.float32)
                add_pair(p0, p1)
    pos_a = torch.tensor(np.stack(pos), dtype=torch.float32)
    batch_a = torch.tensor(batch, dtype=torch.long)
    ei_a, disp_a = rg(pos_a, torch.zeros(b, 3, 3), batch_a)
    dn = (pos_a[0::2] - pos_a[1::2]).norm(dim=1)
    ties_a = int(((dn - r).abs() <= 4 * np.spacing(np.float32(r))).sum())
    # ---- (b) periodic: one cell per molecule (every pair has its own cell entry in the batch)
    pos, batch, cells = [], [], []
    b = 0
    cell_list = [np.diag([10.0, 10.0, 10.0]), np.diag([10.0, 13.0, 11.0]),
                 np.array([[10.0, 0, 0], [3.0, np.sqrt(100.0 - 9.0), 0], [0, 0, 12.0]]),
                 np.array([[10.0, 0, 0], [2.0, 11.0, 0], [-6.0, 0.0, 8.0]])]
    for c in cell_list:
        c32 = c.astype(np.float32)
        for axis in range(3):
            a = c32[axis].astype(np.float64)
            for sgn in (1.0, -1.0):
                for k in (-4, -2, -1, 0, 1, 2, 4):
                    for perp in (0.0, 1e-4, 0.3):
                        for off in (False, True):
                            o = (torch.rand(3, generator=g, dtype=torch.float64).numpy() * 2.0) if off else np.zeros(3)
                            half = np.float64(_ulp_steps(0.5, k))
                            e = np.zeros(3)
                            e[(axis + 1) % 3] = perp
                            p0 = o.astype(np.float32)
                            p1 = (p0.astype(np.float64) + sgn * half * a + e).astype(np.float32)
                            pos.extend([p0, p1])
                            batch.extend([b, b])
                            cells.append(c32)
                            b += 1
    pos_b = torch.tensor(np.stack(pos), dtype=torch.float32)
    batch_b = torch.tensor(batch, dtype=torch.long)
    cell_b = torch.tensor(np.stack(cells), dtype=torch.float32)
    ei_b, disp_b = rg(pos_b, cell_b, batch_b)
    np.savez_compressed(f'{OUT}/case_boundary.npz', cutoff=r,
                        a_pos=pos_a.numpy(), a_batch=batch_a.numpy(), a_edge_index=ei_a.numpy(), a_disp=disp_a.numpy(),
                        b_pos=pos_b.numpy(), b_batch=batch_b.numpy(), b_cell=cell_b.numpy(), b_edge_index=ei_b.numpy(),
                        b_disp=disp_b.numpy())
    print('boundary (a):', pos_a.shape[0] // 2, 'pairs,', ei_a.shape[1] // 2, 'inside;', ties_a, 'within 4 ulp of r')
    print('boundary (b):', pos_b.shape[0] // 2, 'pairs,', ei_b.shape[1] // 2, 'inside')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'virial':
    main_virial()
if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'boundary':
    main_boundary()


def main_cosine():
    """Reference run with the edge embedding's envelope swapped for CosineCutoff (representations.py:177-203) and, second,
    PolynomialCutoff(p=6): aspirin8 geometry, seeded weights, ['energy', 'gradient_force'] -> case_envelope.npz."""
    NewtonNet = import_reference()
    from newtonnet.layers.representations import CosineCutoff, PolynomialCutoff
    rnd = {k: torch.from_numpy(v).double() for k, v in np.load(f'{OUT}/rand_state_seed0.npz').items()}
    train = read_extxyz(f'{REF}/scripts/md17_data/aspirin/ccsd_train/raw/aspirin_ccsd-train.xyz', 8)
    z = torch.tensor(train[0][0], dtype=torch.long).repeat(8)
    pos = torch.tensor(np.concatenate([f[1] for f in train]))
    batch = torch.repeat_interleave(torch.arange(8), 21)
    cell = torch.zeros(8, 3, 3, dtype=torch.float64)
    rec = dict(z=z.numpy(), pos=pos.numpy(), cell=cell.numpy(), batch=batch.numpy())
    for tag, env in (('cosine', CosineCutoff()), ('poly6', PolynomialCutoff(6))):
        model = NewtonNet(output_properties=['energy', 'gradient_force'])
        model.to(torch.float64)
        model.load_state_dict(rnd, strict=True)
        model.embedding_layers.edge_embedding.envelope = env
        model.eval()
        edge = {}
        h = model.embedding_layers.edge_embedding.register_forward_hook(lambda m, i, o: edge.update(dist_edge=o[0].detach().clone()))
        out = model(z, pos.clone(), cell, batch)
        h.remove()
        rec[f'{tag}_energy'] = out.energy.detach().numpy()
        rec[f'{tag}_forces'] = out.gradient_force.detach().numpy()
        rec[f'{tag}_dist_edge'] = edge['dist_edge'].numpy()
        rec['edge_index'] = out.edge_index.numpy()
        print('envelope', tag, rec[f'{tag}_energy'][:2], float(np.abs(rec[f'{tag}_forces']).max()))
    np.savez_compressed(f'{OUT}/case_envelope.npz', **rec)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'cosine':
    main_cosine()


def main_train():
    """The reference's training objective and its parameter gradients (train/trainer.py:299-313): the reference model in train
    mode (derivative heads with create_graph, models/newtonnet.py:106-113), seeded weights, the mixed-molecule case, the loss
    factory of the reference itself with the published weights (train/loss.py:5-52, scripts/config.yml:45-51: MSE(E) + 50 MSE(F)),
    seeded labels, loss.backward() -> case_train_mixed.npz (loss + every parameter gradient, float64 run stored as float32)."""
    import types as _types
    NewtonNet = import_reference()
    import importlib.util
    spec = importlib.util.spec_from_file_location('_ref_loss', f'{REF}/newtonnet/train/loss.py')   # (the package __init__ pulls in wandb)
    _ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(_ref_loss)
    get_loss_by_string = _ref_loss.get_loss_by_string
    rnd = {k: torch.from_numpy(v).double() for k, v in np.load(f'{OUT}/rand_state_seed0.npz').items()}
    c = np.load(f'{OUT}/case_mixed_rand.npz')
    z, pos = torch.from_numpy(c['z']).long(), torch.from_numpy(c['pos']).double()
    cell, batch = torch.from_numpy(c['cell']).double(), torch.from_numpy(c['batch']).long()
    g = torch.Generator().manual_seed(3)
    e_lab = torch.randn(cell.shape[0], generator=g)            # float32 draws (the GPU tests draw the same), widened
    f_lab = torch.randn(pos.shape[0], 3, generator=g)
    model = NewtonNet(output_properties=['energy', 'gradient_force'])
    model.to(torch.float64)
    model.load_state_dict(rnd, strict=True)
    model.train()
    main_loss, _ = get_loss_by_string({'energy': {'weight': 1.0, 'mode': 'mse'}, 'gradient_force': {'weight': 50.0, 'mode': 'mse'}})
    data = _types.SimpleNamespace(z=z, batch=batch, energy=e_lab.double(), force=f_lab.double())
    pred = model(z, pos.clone(), cell, batch)
    loss = main_loss(pred, data)
    loss.backward()
    rec = dict(loss=np.float64(loss.item()), energy_label=e_lab.numpy(), force_label=f_lab.numpy(),
               energy=pred.energy.detach().numpy(), forces=pred.gradient_force.detach().numpy())
    n = 0
    for name, p in model.named_parameters():
        if p.grad is not None:
            rec['grad.' + name] = p.grad.detach().numpy().astype(np.float32)
            rec['gnorm.' + name] = np.float64(p.grad.detach().norm().item())
            n += 1
    print('train fixture: loss', loss.item(), n, 'parameter gradients')
    np.savez_compressed(f'{OUT}/case_train_mixed.npz', **rec)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'train':
    main_train()


def main_triclinic_fuzz(n=100000, seed=2024):
    """Expected neighbor bits of the triclinic fuzz (tests/util.py:triclinic_fuzz_inputs): the reference's own RadiusGraph in
    fp32 on n two-atom cells, run in chunks (its per-molecule Python loop is O(B N)).  The fixture stores ONLY the expected
    bits (is 0 -> 1 an edge, is 1 -> 0 an edge) and a checksum of the regenerated inputs; the inputs come from the seed."""
    import hashlib
    import_reference()
    from newtonnet.layers.representations import RadiusGraph
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from tests import util
    r = 5.0
    rg = RadiusGraph(r)
    pos, cells, batch, kinds = util.triclinic_fuzz_inputs(n, seed, r)
    bits = np.zeros((n, 2), dtype=np.uint8)
    chunk = 1000
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        p = torch.from_numpy(pos[2 * c0:2 * c1])
        ei, _ = rg(p, torch.from_numpy(cells[c0:c1]), torch.from_numpy(batch[2 * c0:2 * c1] - c0))
        ei = ei.numpy()
        for i, j in ei.T:
            bits[c0 + i // 2, i % 2] = 1
    digest = hashlib.sha256(pos.tobytes() + cells.tobytes()).hexdigest()
    np.savez_compressed(f'{OUT}/case_triclinic_fuzz.npz', n=n, seed=seed, cutoff=r, bits=np.packbits(bits.reshape(-1)),
                        input_sha256=np.array(digest), kinds=np.bincount(kinds, minlength=3))
    print('triclinic fuzz:', n, 'cells,', int(bits[:, 0].sum()), 'edges 0->1,', int(bits[:, 1].sum()), 'edges 1->0,',
          int((bits[:, 0] != bits[:, 1]).sum()), 'asymmetric; kinds', np.bincount(kinds, minlength=3))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'triclinic_fuzz':
    main_triclinic_fuzz()


def main_train_direct():
    """The reference's training objective for a model WITHOUT a derivative head (trainer.py:299-313, loss.py:30-47): output
    properties ['energy', 'direct_force'], its own loss factory {'energy': mse x 1, 'direct_force': mse x 20}, loss.backward().
    Shared parameters = rand_state_seed0.npz; the direct_force head and its scaler are drawn by the reference's own constructor
    under torch.manual_seed(5) and stored (float32) next to the loss and every parameter gradient -> case_train_direct.npz."""
    import types as _types
    NewtonNet = import_reference()
    import importlib.util
    spec = importlib.util.spec_from_file_location('_ref_loss', f'{REF}/newtonnet/train/loss.py')
    _ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(_ref_loss)
    rnd = {k: torch.from_numpy(v).double() for k, v in np.load(f'{OUT}/rand_state_seed0.npz').items()}
    c = np.load(f'{OUT}/case_mixed_rand.npz')
    z, pos = torch.from_numpy(c['z']).long(), torch.from_numpy(c['pos']).double()
    cell, batch = torch.from_numpy(c['cell']).double(), torch.from_numpy(c['batch']).long()
    g = torch.Generator().manual_seed(3)
    e_lab = torch.randn(cell.shape[0], generator=g)
    f_lab = torch.randn(pos.shape[0], 3, generator=g)
    torch.manual_seed(5)
    model = NewtonNet(output_properties=['energy', 'direct_force'])
    with torch.no_grad():      # a non-trivial per-element scale (the constructor's is all ones)
        model.scalers[1].scale.weight.copy_(1.0 + 0.25 * torch.randn(model.scalers[1].scale.weight.shape))
    head_sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()
               if k.startswith('output_layers.1.') or k.startswith('scalers.1.')}
    sd = {k: (rnd[k] if k in rnd else head_sd[k].double()) for k in model.state_dict().keys()}
    model.to(torch.float64)
    model.load_state_dict(sd, strict=True)
    model.train()
    main_loss, _ = _ref_loss.get_loss_by_string({'energy': {'weight': 1.0, 'mode': 'mse'},
                                                 'direct_force': {'weight': 20.0, 'mode': 'mse'}})
    data = _types.SimpleNamespace(z=z, batch=batch, energy=e_lab.double(), force=f_lab.double())
    pred = model(z, pos.clone(), cell, batch)
    loss = main_loss(pred, data)
    loss.backward()
    rec = dict(loss=np.float64(loss.item()), energy_label=e_lab.numpy(), force_label=f_lab.numpy(),
               energy=pred.energy.detach().numpy(), direct_force=pred.direct_force.detach().numpy())
    for k, v in head_sd.items():
        rec['state.' + k] = v.numpy()
    n = 0
    for name, p in model.named_parameters():
        if p.grad is not None:
            rec['grad.' + name] = p.grad.detach().numpy().astype(np.float32)
            n += 1
    print('train-direct fixture: loss', loss.item(), n, 'parameter gradients')
    np.savez_compressed(f'{OUT}/case_train_direct.npz', **rec)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'train_direct':
    main_train_direct()
