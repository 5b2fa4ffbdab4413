#!/usr/bin/env python3
"""us per step of the row path and the fused edge phase (csrc/molfuse2.hip) at fixed molecule COUNTS for molecules of different sizes
(the MD17 shapes of bench.synthetic_md17_mixed, one shape at a time, and their mix).  MEDIAN of five regions of 30 steps per form, the
two forms interleaved region by region (round 5's file had first-shape outliers from single regions).
usage: python tools/fused_by_molecule_size.py [B ...]"""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()


def ab(args, reps=30, regions=5):
    f, t = {}, {0: [], 1: []}
    for m in (0, 1):
        hip.set_mol_fused(m)
        for _ in range(8):
            out = model(*args)
        f[m] = out.gradient_force.clone()
    torch.cuda.synchronize()
    for _ in range(regions):
        for m in (0, 1):
            hip.set_mol_fused(m)
            out = model(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = model(*args)
            torch.cuda.synchronize()
            t[m].append((time.perf_counter() - t0) / reps * 1e6)
    return statistics.median(t[0]), statistics.median(t[1]), float((f[0] - f[1]).abs().max())


for B in [int(a) for a in sys.argv[1:]] or [448, 512, 1024]:
    z, pos, cell, batch = bench.synthetic_md17_mixed(9 * B, 0, 'cuda')[:4]
    sizes = torch.bincount(batch)
    print(f'B = {B}', flush=True)
    for k, (name, _) in enumerate(bench.MD17_SHAPES):
        mols = torch.arange(k, 9 * B, 9)[:B]                     # B molecules of shape k
        keep = torch.isin(batch.cpu(), mols).cuda()
        zz, pp = z[keep], pos[keep]
        bb = torch.repeat_interleave(torch.arange(B, device='cuda'), int(sizes[k]))
        t0, t1, d = ab((zz, pp, torch.zeros(B, 3, 3, device='cuda'), bb))
        print(f'  {name:14s} {int(sizes[k]):2d} atoms: row path {t0:7.1f} us, fused {t1:7.1f} us ({100 * (t1 / t0 - 1):+.1f} %), max |dF| {d:.1e}', flush=True)
    t0, t1, d = ab(bench.synthetic_md17_mixed(B, 0, 'cuda')[:4])
    print(f'  mix of the nine      : row path {t0:7.1f} us, fused {t1:7.1f} us ({100 * (t1 / t0 - 1):+.1f} %), max |dF| {d:.1e}', flush=True)
hip.set_mol_fused(-1)
