#!/usr/bin/env python3
"""Experiment: split the config-2 batch into K molecule chunks on K HIP streams, so that one chunk's HBM-bound edge kernels can
overlap another chunk's MFMA-bound MLP kernels.  Static candidate lists (refresh_graph) keep the host out of it: no edge-count
sync, so the streams really run side by side."""
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
prep = hip.prepare(m, 'cuda')
for K in (1, 2, 4):
    B = 1024 // K
    streams = [torch.cuda.Stream() for _ in range(K)]
    sl = [(k * B * 21, (k + 1) * B * 21, k * B, (k + 1) * B) for k in range(K)]
    zs = [z[a0:a1].contiguous() for a0, a1, _, _ in sl]
    ps = [pos[a0:a1].contiguous() for a0, a1, _, _ in sl]
    cs = [cell[b0:b1].contiguous() for _, _, b0, b1 in sl]
    bs = [(batch[a0:a1] - b0).contiguous() for a0, a1, b0, _ in sl]
    gs = [hip.build_graph(ps[k], cs[k], bs[k], 5.0, freq) for k in range(K)]
    outs = [hip.energy_forces(m, zs[k], ps[k], cs[k], gs[k], want_nodes=False, prepared=prep) for k in range(K)]
    ws = [o.pop('workspace') for o in outs]
    torch.cuda.synchronize()
    def step():
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                hip.refresh_graph(gs[k], ps[k], cs[k], bs[k], 5.0, freq)
                hip.energy_forces(m, zs[k], ps[k], cs[k], gs[k], want_nodes=False, workspace=ws[k], out=outs[k], prepared=prep)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f'K={K}: {dt*1e3:.3f} ms/step  {N/dt/1e6:.2f} M atom-steps/s', flush=True)
