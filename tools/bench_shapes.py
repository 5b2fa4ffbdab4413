#!/usr/bin/env python3
"""Throughput sanity across molecule shapes (same kernels, different row lengths / molecule sizes): synthetic batches of
n-atom molecules at liquid-like density, ~21.5k atoms each.   usage: python tools/bench_shapes.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
rng = np.random.default_rng(0)
for n in (3, 9, 21, 32, 64, 128):
    B = max(1, 21504 // n)
    box = (n * 14.0) ** (1 / 3)
    # jittered lattice: no close contacts
    m = int(np.ceil(n ** (1 / 3)))
    grid = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing='ij'), -1).reshape(-1, 3)[:n] * (box / m)
    pos = np.concatenate([grid + rng.normal(0, 0.12, grid.shape) for _ in range(B)])
    z = torch.tensor(rng.choice([1, 6, 8], n * B), dtype=torch.long, device='cuda')
    pos = torch.tensor(pos, dtype=torch.float32, device='cuda')
    cell = torch.zeros(B, 3, 3, device='cuda')
    batch = torch.arange(B, device='cuda').repeat_interleave(n)
    for _ in range(5): out = model(z, pos, cell, batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = model(z, pos, cell, batch)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    E = out.edge_index.shape[1]
    print(f'n={n:4d} B={B:5d} N={n*B:6d} E={E:8d} ({E/(n*B):5.1f}/atom): {dt*1e3:7.3f} ms/step  {n*B/dt/1e6:6.2f} M atom-steps/s  '
          f'{E/dt/1e6:7.1f} M edge-steps/s', flush=True)
