#!/usr/bin/env python3
"""Config 5 (100k-atom periodic box): what would an internal spatial re-ordering of the atoms buy?  Measured the cheap way: the
INPUT is permuted on the host (recipe order = lattice raster; cell-major over 5 A cells; Morton order of those cells; random)
and the unchanged library timed on each.  The step's results are permutation-equivariant, so the time is what an internal
permutation could reach (minus its own gather / scatter).
usage: python tools/box_order_ab.py [n_atoms]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
z, pos, cell, batch = bench.synthetic_box(n, 47, 0, 'cpu')


def part1by2(v):
    v = v.astype(np.uint64) & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


c = np.floor(pos.numpy() / 5.0).astype(np.int64) % 20
orders = {'recipe (lattice raster)': np.arange(n),
          'cell-major (5 A cells, x slowest)': np.lexsort((np.arange(n), c[:, 2], c[:, 1], c[:, 0])),
          'Morton order of the 5 A cells': np.argsort((part1by2(c[:, 0]) << 2) | (part1by2(c[:, 1]) << 1) | part1by2(c[:, 2]), kind='stable'),
          'random': np.random.default_rng(0).permutation(n)}
ref = None
names = list(orders)
if os.environ.get('BOX_ORDER_REVERSED') == '1':
    names.reverse()
for name in names:
    perm = orders[name]
    p = torch.from_numpy(perm)
    args = (z[p].cuda(), pos[p].cuda(), cell.cuda(), batch.cuda())
    for _ in range(4):
        out = model(*args)
        f = out.gradient_force
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 12
    for _ in range(reps):
        out = model(*args)
        f = out.gradient_force
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    fu = torch.empty_like(f.cpu())
    fu[p] = f.cpu()
    if ref is None:
        ref = fu
    print(f'{name:40s} {ms:7.2f} ms per step ({n / ms * 1e-3:.2f} M atom-steps/s); max |dF| vs the recipe order {(fu - ref).abs().max():.1e}', flush=True)
