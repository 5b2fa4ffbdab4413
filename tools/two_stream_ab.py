#!/usr/bin/env python3
"""Would two half-batches on two streams shorten a small step?  One module at B conformers against two modules (same weights) at
B/2 each, queued alternately from one host thread on two streams (the launch chains of the halves can overlap on the GPU).
usage: python tools/two_stream_ab.py [B ...]"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
m0 = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); m0.eval()
ms = [copy.deepcopy(m0) for _ in range(4)]
for m in ms:
    m.eval()
streams = [torch.cuda.Stream() for _ in range(4)]
for B in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]:
    full = bench.synthetic_aspirin(B, 0, 'cuda')
    res = {}
    for parts in (1, 2, 4):
        if B % parts:
            continue
        n = 21 * (B // parts)
        shards = [(full[0][k * n:(k + 1) * n].clone(), full[1][k * n:(k + 1) * n].clone(), full[2][:B // parts].clone(),
                   full[3][:n].clone()) for k in range(parts)]

        def step():
            outs = []
            for k in range(parts):
                with torch.cuda.stream(streams[k]):
                    outs.append(ms[k](*shards[k]))
            return outs
        for _ in range(20):
            outs = step()
        for o in outs:
            o.gradient_force
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            outs = step()
        for k, o in enumerate(outs):
            with torch.cuda.stream(streams[k]):
                o.gradient_force
        torch.cuda.synchronize()
        res[parts] = (time.perf_counter() - t0) / reps * 1e6
    print(f'B = {B}: one stream {res[1]:.0f} us per step' + ''.join(f'; {p} streams x {B // p} conformers {res[p]:.0f} us' for p in (2, 4) if p in res), flush=True)
