#!/usr/bin/env python3
"""us per step of the row path (NNHIP_MOL_FUSED=0) and molfuse2.hip (=6) at fixed molecule COUNTS for molecules of different sizes
(the MD17 shapes of bench.synthetic_md17_mixed, one shape at a time, and their mix).
usage: python tools/fused_by_molecule_size.py [B ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()


def timed(args, mode, reps=60):
    os.environ['NNHIP_MOL_FUSED'] = mode
    for _ in range(8):
        out = model(*args)
    f = out.gradient_force.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = model(*args)
    out.gradient_force
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, f


for B in [int(a) for a in sys.argv[1:]] or [448, 512]:
    z, pos, cell, batch = bench.synthetic_md17_mixed(9 * B, 0, 'cuda')[:4]
    sizes = torch.bincount(batch)
    print(f'B = {B}', flush=True)
    for k, (name, _) in enumerate(bench.MD17_SHAPES):
        mols = torch.arange(k, 9 * B, 9)[:B]                     # B molecules of shape k
        keep = torch.isin(batch.cpu(), mols).cuda()
        zz, pp = z[keep], pos[keep]
        bb = torch.repeat_interleave(torch.arange(B, device='cuda'), int(sizes[k]))
        args = (zz, pp, torch.zeros(B, 3, 3, device='cuda'), bb)
        t0, f0 = timed(args, '0')
        t6, f6 = timed(args, '6')
        print(f'  {name:14s} {int(sizes[k]):2d} atoms: row path {t0:7.1f} us, fused {t6:7.1f} us ({100 * (t6 / t0 - 1):+.1f} %), max |dF| {float((f0 - f6).abs().max()):.1e}', flush=True)
    args = bench.synthetic_md17_mixed(B, 0, 'cuda')[:4]
    t0, f0 = timed(args, '0')
    t6, f6 = timed(args, '6')
    print(f'  mix of the nine      : row path {t0:7.1f} us, fused {t6:7.1f} us ({100 * (t6 / t0 - 1):+.1f} %), max |dF| {float((f0 - f6).abs().max()):.1e}', flush=True)
