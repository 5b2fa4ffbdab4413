#!/usr/bin/env python3
"""One-off stress of the persistent (large-batch) kernels on irregular batches: a few hundred random molecules of 8-40 atoms
(ragged rows, > 27 k pair rows: the persistent split-f16 edge-MLP kernel) against the float64 oracle on the host.
usage: python tools/fuzz_large.py [n_molecules] [n_seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import newtonnet_ref as ref  # noqa: E402   (tooling: a checker, like the tests)
from tests import util  # noqa: E402
from tests.test_hip_parity import make_model  # noqa: E402

n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 350
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    rng = np.random.default_rng(7000 + seed)
    model, sd = make_model('rand' if seed % 2 == 0 else 'ckpt')
    zs, ps = [], []
    for n in rng.integers(8, 41, size=n_mol):
        box = (n * rng.uniform(15.0, 40.0)) ** (1.0 / 3.0)
        p = rng.uniform(0, box, size=(n, 3))
        for _ in range(200):
            d = np.linalg.norm(p[:, None] - p[None], axis=-1) + np.eye(n) * 9
            bad = np.argwhere(d < 0.9)
            if len(bad) == 0:
                break
            p[bad[:, 0]] = rng.uniform(0, box, size=(len(bad), 3))
        else:
            continue
        zs.append(rng.choice([1, 6, 7, 8], n))
        ps.append(p)
    z = torch.tensor(np.concatenate(zs), dtype=torch.long)
    pos = torch.tensor(np.concatenate(ps), dtype=torch.float32)
    cell = torch.zeros(len(zs), 3, 3)
    batch = torch.tensor(np.concatenate([[b] * len(q) for b, q in enumerate(zs)]), dtype=torch.long)
    out = model(z.cuda(), pos.cuda(), cell.cuda(), batch.cuda())
    again = model(z.cuda(), pos.cuda(), cell.cuda(), batch.cuda())      # the deferred (steady-state) path: same bits
    assert torch.equal(again.edge_index, out.edge_index) and torch.equal(again.energy, out.energy)
    assert torch.equal(again.gradient_force, out.gradient_force) and torch.equal(again.atom_node, out.atom_node)
    want = ref.energy_forces({k: v.double() for k, v in sd.items()}, z, pos.double(), cell.double(), batch)
    assert np.array_equal(out.edge_index.cpu().numpy(), want['edge_index'].numpy()), 'edge_index differs'
    e, f = want['energy'].numpy(), want['forces'].numpy()
    de = np.abs(out.energy.cpu().numpy() - e)
    df = np.abs(out.gradient_force.cpu().numpy() - f)
    scale = max(1.0, float(np.abs(f).max()))
    print(f'seed {seed}: N {len(z)} pairs {out.edge_index.shape[1] // 2} max|F| {np.abs(f).max():.3e}  force MAE {df.mean():.2e} '
          f'max {df.max():.2e} (x{scale:.1e})  max |dE| {de.max():.2e} of {np.abs(e).max():.2e}', flush=True)
    ok = df.mean() <= util.FORCE_MAE_TOL * scale and df.max() <= util.FORCE_MAX_TOL * scale and np.all(de <= util.energy_tol(e) + 2e-6 * np.abs(e).max())
    print('   within the stated tolerances' if ok else '   OUTSIDE the stated tolerances (see the magnitudes)')
print('ok')
