#!/usr/bin/env python3
"""Experiment: split the config-2 batch into K molecule chunks and run them on K HIP streams, so that one chunk's
HBM-bound edge kernels can overlap another chunk's MFMA-bound MLP kernels."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet

torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
z, pos, cell, batch = bench.synthetic_aspirin(1024, 0, 'cuda')
N = z.shape[0]
freq = model.embedding_layers.edge_embedding.embedding.frequencies
m = model._hip_model(0)

def run_chunks(K, streams, ws):
    B = 1024 // K
    outs = []
    for k in range(K):
        with torch.cuda.stream(streams[k]):
            a0, a1 = k * B * 21, (k + 1) * B * 21
            g = hip.build_graph(pos[a0:a1], cell[k * B:(k + 1) * B], batch[a0:a1] - k * B, 5.0, freq)
            outs.append(hip.energy_forces(m, z[a0:a1], pos[a0:a1], cell[k * B:(k + 1) * B], g, want_nodes=False, workspace=ws[k]))
    return outs

for K in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(K)]
    ws = [None] * K
    outs = run_chunks(K, streams, ws)
    ws = [o['workspace'] for o in outs]
    torch.cuda.synchronize()
    for _ in range(3): run_chunks(K, streams, ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run_chunks(K, streams, ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f'K={K}: {dt*1e3:.3f} ms/step  {N/dt/1e6:.2f} M atom-steps/s', flush=True)
