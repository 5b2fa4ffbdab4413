#!/usr/bin/env python3
"""Soak of `model.inference_lanes`: three lanes on three streams, every step of every lane compared bitwise (on the device, on the
lane's own stream) with the module's one-stream result for that batch -- batches of several shapes cycling through the lanes so that
consecutive in-flight steps differ in size, the comparison results collected at the end (no host sync inside the run: the steps really
overlap).  usage: python tools/soak_lanes.py [rounds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.manual_seed(0)
dev = 'cuda'
model = NewtonNet(output_properties=['energy', 'gradient_force']).to(dev)
model.eval()
batches = [bench.synthetic_aspirin(B, s, dev) for B, s in ((1024, 0), (1024, 1), (128, 2), (512, 3), (1024, 4), (64, 5), (700, 6))]
batches.append(bench.synthetic_md17_mixed(288, 0, dev)[:4])
want = []
for b in batches:
    o = model(*b)
    want.append((o.energy.clone(), o.gradient_force.clone()))
torch.cuda.synchronize()
for n_lanes in (2, 3):
    lanes = model.inference_lanes(n_lanes)
    streams = [torch.cuda.Stream() for _ in range(n_lanes)]
    bad = torch.zeros(n_lanes, dtype=torch.int64, device=dev)
    t0, n = time.time(), 0
    for r in range(rounds):
        for k in range(n_lanes):
            b = (r * n_lanes + k * 3 + r // 7) % len(batches)
            with torch.cuda.stream(streams[k]):
                o = lanes[k](*batches[b])
                ok = torch.equal(o.energy, want[b][0]) if False else None      # (torch.equal syncs: compare on the device instead)
                miss = (o.energy != want[b][0]).any() | (o.gradient_force != want[b][1]).any()
                bad[k] += miss.to(torch.int64)
            n += 1
        if r % 8 == 7:
            for s in streams:       # bound how far the host runs ahead (the allocator's per-stream pools stay small)
                s.synchronize()
    torch.cuda.synchronize()
    st = [l.deferred_stats() for l in lanes]
    print(f'{n_lanes} lanes: {n} steps over {len(batches)} batch shapes in {time.time() - t0:.1f} s; steps that differ from the one-stream '
          f'result, per lane: {bad.tolist()}; repeats needed: {[s["repeats_needed"] for s in st]}', flush=True)
