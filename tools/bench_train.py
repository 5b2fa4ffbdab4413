#!/usr/bin/env python3
"""Train-step timing for BASELINE configs[2] (MD17 ethanol-shaped, batch 32 and 10): loss = MSE(E) + 50 MSE(F), Adam,
clip 1.0 -- GPU train-mode path (fp32 and bf16 autocast) vs the CPU oracle's double backward on 16 host threads."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newtonnet_amd.distributed import GraphedTrainStep, TrainStep
from newtonnet_amd.models import NewtonNet
from oracle import newtonnet_ref as ref

def ethanol_batch(B, seed=0):
    eth0 = torch.tensor([[0.00, 0.00, 0.00], [1.52, 0.00, 0.00], [2.05, 1.32, 0.00], [-0.39, 1.02, 0.00],
                         [-0.39, -0.51, 0.89], [-0.39, -0.51, -0.89], [1.90, -0.53, 0.88], [1.90, -0.53, -0.88],
                         [3.01, 1.30, 0.00]])
    g = torch.Generator().manual_seed(seed)
    pos = eth0.repeat(B, 1) + 0.1 * torch.randn(9 * B, 3, generator=g)
    z = torch.tensor([6, 6, 8, 1, 1, 1, 1, 1, 1]).repeat(B)
    batch = torch.repeat_interleave(torch.arange(B), 9)
    return z, pos, torch.zeros(B, 3, 3), batch, torch.randn(B, generator=g), torch.randn(9 * B, 3, generator=g)

for B in (10, 32, 256):
    z, pos, cell, batch, e_lab, f_lab = ethanol_batch(B)
    torch.manual_seed(0)
    model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.train()
    step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), 1.0, 50.0, 1.0)
    args = [t.cuda() for t in (z, pos, cell, batch, e_lab, f_lab)]
    res = {}
    for name, ctx in (('fp32', torch.autocast('cuda', enabled=False)), ('bf16', torch.autocast('cuda', dtype=torch.bfloat16))):
        with ctx:
            for _ in range(5): step(*args)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): step(*args)
            torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 20
    torch.manual_seed(0)
    model_g = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model_g.train()
    gstep = GraphedTrainStep(model_g, torch.optim.Adam(model_g.parameters(), lr=1e-3, capturable=True), 1.0, 50.0, 1.0)
    for _ in range(3): gstep(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): gstep(*args)
    torch.cuda.synchronize(); res['graphed'] = (time.perf_counter() - t0) / 50
    torch.manual_seed(0)
    model_b = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model_b.train()
    bstep = GraphedTrainStep(model_b, torch.optim.Adam(model_b.parameters(), lr=1e-3, capturable=True), 1.0, 50.0, 1.0)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        for _ in range(3): bstep(*args)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): bstep(*args)
        torch.cuda.synchronize(); res['graphed_bf16'] = (time.perf_counter() - t0) / 50
    torch.set_num_threads(min(16, os.cpu_count()))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref.training_loss_grads(sd, z, pos, cell, batch, e_lab, f_lab)
    t0 = time.perf_counter(); n = 3
    for _ in range(n): ref.training_loss_grads(sd, z, pos, cell, batch, e_lab, f_lab)
    cpu = (time.perf_counter() - t0) / n
    print(f'B={B:4d} ({9*B} atoms): GPU train step eager fp32 {res["fp32"]*1e3:7.2f} ms | bf16 {res["bf16"]*1e3:7.2f} ms | '
          f'HIP-graph replay fp32 {res["graphed"]*1e3:6.2f} ms bf16 {res["graphed_bf16"]*1e3:6.2f} ms | CPU oracle loss+grads {cpu*1e3:8.1f} ms '
          f'({torch.get_num_threads()} threads) | speedup eager {cpu/res["fp32"]:.1f}x graphed {cpu/res["graphed"]:.1f}x', flush=True)
