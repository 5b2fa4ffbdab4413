#!/usr/bin/env python3
"""Run a few training steps (fused path, HIP-graph replay unless --eager) for rocprofv3.
usage: tools/profile_train.py [ethanol|aspirin] [B] [--eager | --fused | --fused-eager] [--steps K]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from bench_train import aspirin_batch, ethanol_batch  # noqa: E402
from newtonnet_amd.distributed import FusedClipAdam, GraphedTrainStep, TrainStep  # noqa: E402
from newtonnet_amd.models import NewtonNet  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'ethanol'
B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 32
eager = '--eager' in sys.argv
steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 10
args = [t.cuda() for t in (aspirin_batch if kind == 'aspirin' else ethanol_batch)(B)]
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
model.train()
if '--fused-eager' in sys.argv:       # no autograd, no capture: exact neighbor list every step
    step = TrainStep(model, FusedClipAdam(model, lr=1e-3, max_norm=1.0), 1.0, 50.0)
elif '--fused' in sys.argv:
    step = GraphedTrainStep(model, FusedClipAdam(model, lr=1e-3, max_norm=1.0), 1.0, 50.0, assume_static=True)
elif eager:
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), 1.0, 50.0, 1.0)
else:
    step = GraphedTrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True), 1.0, 50.0, 1.0)
for _ in range(steps):
    loss = step(*args)
torch.cuda.synchronize()
print(kind, B, 'loss', float(loss))
