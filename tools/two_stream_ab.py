#!/usr/bin/env python3
"""Steps in flight.  (a) ONE batch of B conformers split into 2 / 4 parts on their own streams and lanes (model.inference_lanes):
the launch chains of the parts can overlap; (b) WHOLE independent steps alternating over 2 / 3 / 4 lanes: what a throughput-oriented
caller with independent batches does, and what bench.py times with --streams.   usage: python tools/two_stream_ab.py [B ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
m0 = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); m0.eval()
ms = m0.inference_lanes(4)
streams = [torch.cuda.Stream() for _ in range(4)]


def timed(step, reps, n_streams):
    for _ in range(20):
        outs = step()
    for o in outs:
        o.gradient_force
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            outs = step()
        for k, o in enumerate(outs):
            with torch.cuda.stream(streams[k % n_streams]):
                o.gradient_force
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / reps * 1e6)
    return sorted(best)[1]


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
        res[parts] = timed(step, 100, parts)
    fulls = [tuple(t.clone() for t in full) for _ in range(4)]
    alt = {}
    for S in (2, 3, 4):
        def stepS():
            outs = []
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    outs.append(ms[k](*fulls[k]))
            return outs
        alt[S] = timed(stepS, 60, S) / S
    print(f'B = {B}: one stream {res[1]:.0f} us per step' + ''.join(f'; split over {p} streams {res[p]:.0f} us' for p in (2, 4) if p in res)
          + ''.join(f'; whole steps over {S} lanes {alt[S]:.0f} us per step' for S in (2, 3, 4)), flush=True)
