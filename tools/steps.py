#!/usr/bin/env python3
"""Run eval-mode energy+force steps on B aspirin conformers (for profilers).  usage: python tools/steps.py [B] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.eval()
z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
for _ in range(n):
    out = model(z, pos, cell, batch)
out.energy.sum().item()
torch.cuda.synchronize()
