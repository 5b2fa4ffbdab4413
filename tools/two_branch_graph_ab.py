#!/usr/bin/env python3
"""Would a small step run faster as TWO half-batches in the two branches of one captured HIP graph (no host launch cost, the
launch chains of the halves free to overlap on the GPU)?  Proxy: refresh_graph + energy_forces on fixed neighbor lists (the
calculator's MD step), captured (a) once for B conformers, (b) as `parts` branches of B / parts conformers each.
usage: python tools/two_branch_graph_ab.py [B ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
emb = model.embedding_layers.edge_embedding
hm = model._hip_model(0)
prep = hip.prepare(hm, torch.device('cuda'))


def make_part(z, pos, cell, batch):
    g = hip.build_graph(pos, cell, batch, emb.cutoff, emb.embedding.frequencies, envelope=emb.envelope_id)
    n, B = pos.shape[0], cell.shape[0]
    out = hip.alloc_outputs(n, B, pos.device, True, False, False)
    ws = torch.empty(max(hip.lib().nnhip_workspace_bytes(n, g.n_edges, B, hm.n_layers), 256), dtype=torch.uint8, device='cuda')
    def run():
        hip.refresh_graph(g, pos, cell, batch, emb.cutoff, emb.embedding.frequencies)
        hip.energy_forces(hm, z, pos, cell, g, want_forces=True, want_virial=False, want_nodes=False, workspace=ws, out=out, prepared=prep)
    return run, out


for B in [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]:
    full = bench.synthetic_aspirin(B, 0, 'cuda')
    line = f'B = {B}:'
    ref = None
    for parts in (1, 2, 4):
        n = 21 * (B // parts)
        runs = [make_part(full[0][k * n:(k + 1) * n].clone(), full[1][k * n:(k + 1) * n].clone(), full[2][:B // parts].clone(),
                          full[3][:n].clone()) for k in range(parts)]
        side = [torch.cuda.Stream() for _ in range(parts)]
        def step():
            cur = torch.cuda.current_stream()
            for k, (run, _) in enumerate(runs):
                if k == 0:
                    run()
                else:
                    side[k].wait_stream(cur)
                    with torch.cuda.stream(side[k]):
                        run()
            for k in range(1, parts):
                cur.wait_stream(side[k])
        s0 = torch.cuda.Stream()
        s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            step()
        for _ in range(20):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 300
        for _ in range(reps):
            graph.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        f = torch.cat([o['forces'] for _, o in runs])
        if ref is None:
            ref = f.clone()
        line += f'  {parts} branch{"es" if parts > 1 else ""} {us:.0f} us (max |dF| vs one {float((f - ref).abs().max()):.1e})'
    print(line, flush=True)
