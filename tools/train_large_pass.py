#!/usr/bin/env python3
"""The large-batch training pass bench.py:train_roofline times, as a program of its own for rocprofv3 (tools/profile_train_large.sh)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
print(json.dumps(bench.train_roofline(torch.device('cuda', 0), reps=reps)))
