#!/usr/bin/env python3
"""Thread-count sweep of the CPU oracle on the GPU box's host cores (choosing the cpu_baseline setting)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import newtonnet_ref as ref
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = np.load('tests/golden/aspirin_frames.npz')
g = torch.Generator().manual_seed(0)
pos = torch.from_numpy(f['test0_pos']).float().repeat(B, 1) + 0.05 * torch.randn(B * 21, 3, generator=g)
z = torch.from_numpy(f['z']).long().repeat(B)
batch = torch.repeat_interleave(torch.arange(B), 21)
cell = torch.zeros(B, 3, 3)
sd = ref.random_state()
print('cores', os.cpu_count())
for th in [int(a) for a in sys.argv[2:]] or [4, 8, 16, 32, 64]:
    torch.set_num_threads(th)
    ref.energy_forces(sd, z, pos, cell, batch)
    t = time.perf_counter(); ref.energy_forces(sd, z, pos, cell, batch); dt = time.perf_counter() - t
    print(f'B={B} threads={th}: {dt:.2f} s  -> {B*21/dt:.0f} atom-steps/s', flush=True)
