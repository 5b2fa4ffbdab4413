"""Tooling: per-stage wall-clock / shader-clock stamps of node_fwd_split_kernel on a one-molecule step (build the library with
`NNHIP_LIB_NAME=libnewtonnet_hip_nsclk.so csrc/build.sh -DNS_CLOCK_DEBUG`, run with NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=...)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
z, pos, cell, batch = bench.synthetic_aspirin(1, 0, 'cuda')
for _ in range(12):
    out = model(z, pos, cell, batch); f = out.gradient_force.cpu()
torch.cuda.synchronize()
