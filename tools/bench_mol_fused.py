#!/usr/bin/env python3
"""Same-box A/B of the molecule-resident fused edge phase (csrc/molfuse.hip) against the row path across batch sizes:
aspirin conformers, eval-mode energy + forces, back-to-back deferred steps.  NNHIP_MOL_FUSED is read per call, so one process
times every form on the same lease.   usage: python tools/bench_mol_fused.py [B ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet

sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 48, 128, 256, 512, 1024, 2048]
modes = os.environ.get('AB_MODES', '0,1').split(',')     # 0 row path, 1 molfuse both ways, 2 / 3 one direction, 4 molfuse2 forward + row adjoint, 5 molfuse2 forward + molfuse adjoint
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
print('device', torch.cuda.get_device_name(0), 'lib', hip.lib().nnhip_version(), flush=True)
for B in sizes:
    z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
    line = f'B={B:5d} N={21 * B:6d}:'
    ref = None
    for mode in modes:
        os.environ['NNHIP_MOL_FUSED'] = mode
        for _ in range(6):
            out = model(z, pos, cell, batch)
        f = out.gradient_force.clone()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                out = model(z, pos, cell, batch)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        if ref is None:
            ref = f
        line += f'  mode {mode}: {best * 1e6:8.1f} us ({21 * B / best / 1e6:6.2f} M at-st/s, dF vs first {float((f - ref).abs().max()):.1e})'
    print(line, flush=True)
os.environ.pop('NNHIP_MOL_FUSED', None)
