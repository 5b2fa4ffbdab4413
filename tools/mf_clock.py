#!/usr/bin/env python3
"""Per-phase wall-clock stamps of the fused molecule kernels (tooling library built with -DMF_CLOCK_DEBUG).
usage: NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=libnewtonnet_hip_dbg.so python tools/mf_clock.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
os.environ.setdefault('NNHIP_MOL_FUSED', '1')
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
for _ in range(3):
    out = model(z, pos, cell, batch)
    torch.cuda.synchronize()
    print('---- step', flush=True)
