#!/usr/bin/env python3
"""Same-box A/B of the fused edge phase (csrc/molfuse2.hip) against the row path across batch sizes: aspirin conformers, eval-mode
energy + forces, back-to-back deferred steps; MEDIAN over five timed regions of 20 steps per form, forms interleaved region by region
so that a clock drift of the box lands on both.   usage: python tools/bench_mol_fused.py [B ...]   (AB_MODES=0,1,2,3)"""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet

sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 384, 448, 512, 640, 768, 1024, 1536, 2048, 4096]
modes = [int(m) for m in os.environ.get('AB_MODES', '0,1').split(',')]     # 0 row path, 1 fused both ways, 2 / 3 one direction
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
print('device', torch.cuda.get_device_name(0), 'lib', hip.lib().nnhip_version(), flush=True)
for B in sizes:
    z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
    forces, times = {}, {m: [] for m in modes}
    for m in modes:
        hip.set_mol_fused(m)
        for _ in range(6):
            out = model(z, pos, cell, batch)
        forces[m] = out.gradient_force.clone()
    torch.cuda.synchronize()
    for _ in range(5):
        for m in modes:
            hip.set_mol_fused(m)
            out = model(z, pos, cell, batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                out = model(z, pos, cell, batch)
            torch.cuda.synchronize()
            times[m].append((time.perf_counter() - t0) / 20)
    line = f'B={B:5d} N={21 * B:6d}:'
    for m in modes:
        med = statistics.median(times[m])
        line += (f'  mode {m}: {med * 1e6:8.1f} us [{min(times[m]) * 1e6:.0f}..{max(times[m]) * 1e6:.0f}]'
                 f' ({21 * B / med / 1e6:6.2f} M, dF {float((forces[m] - forces[modes[0]]).abs().max()):.1e})')
    if len(modes) > 1:
        line += f'  fused/row {statistics.median(times[modes[1]]) / statistics.median(times[modes[0]]):.3f}'
    print(line, flush=True)
hip.set_mol_fused(-1)
