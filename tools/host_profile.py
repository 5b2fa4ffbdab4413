#!/usr/bin/env python3
"""Where the HOST time of one small eval-mode call goes: cProfile of model() at B = 1 with the GPU kept
out of the way (the deferred path never waits for it), then wall time per call with and without reading the forces back."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet

torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
for _ in range(50):
    model(z, pos, cell, batch).energy
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    out = model(z, pos, cell, batch)
t_queue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    out = model(z, pos, cell, batch)
torch.cuda.synchronize()
t_async = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    f = model(z, pos, cell, batch).gradient_force.cpu()
t_sync = (time.perf_counter() - t0) / n
print(f'B={B}: host time per call (queue only) {1e6 * t_queue:.1f} us; back-to-back calls {1e6 * t_async:.1f} us/step; '
      f'with forces read back every step {1e6 * t_sync:.1f} us/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    out = model(z, pos, cell, batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
