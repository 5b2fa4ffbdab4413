#!/usr/bin/env python3
"""Single-molecule (MD-loop) latency of the calculator path: config 1 shape (21 atoms, 306 edges)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
for B in (1, 8, 64):
    z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
    for _ in range(20): out = model(z, pos, cell, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        out = model(z, pos, cell, batch)
        f = out.gradient_force.cpu()          # an MD driver needs the forces on the host every step
    dt = (time.perf_counter() - t0) / n
    # breakdown
    freq = model.embedding_layers.edge_embedding.embedding.frequencies
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g = hip.build_graph(pos, cell, batch, 5.0, freq)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n
    m = model._hip_model(0); ws = None
    r = hip.energy_forces(m, z, pos, cell, g); ws = r['workspace']
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = hip.energy_forces(m, z, pos, cell, g, workspace=ws)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): m = model._hip_model(0)
    tm = (time.perf_counter() - t0) / n
    print(f'B={B:3d}: full step {dt*1e6:8.1f} us | build_graph {tg*1e6:7.1f} | energy_forces {te*1e6:7.1f} | _hip_model {tm*1e6:6.1f}', flush=True)

# ---- the calculator's MD-loop path (one structure per calculate(), positions change a little every step)
from newtonnet_amd.utils import MLAseCalculator
from tests.test_ase_calculator import FakeAtoms
z, pos, cell, batch = bench.synthetic_aspirin(1, 0, 'cpu')
numbers, p0 = z.numpy(), pos.double().numpy()
rng = np.random.default_rng(0)
vel = rng.normal(0, 0.004, p0.shape)
for label, kw in (('exact list every step (skin=0)', dict(skin=0.0)), ('skin list, direct launches', dict(skin=0.5, capture=False)),
                  ('skin list + HIP graph replay', dict(skin=0.5, capture=True))):
    calc = MLAseCalculator(model, properties=['energy', 'forces'], device='cuda', **kw)
    for s in range(30): calc.calculate(FakeAtoms(numbers, p0 + s * vel))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 400
    for s in range(n): calc.calculate(FakeAtoms(numbers, p0 + (30 + s) * vel))
    dt = (time.perf_counter() - t0) / n
    print(f'calculator, {label:34s}: {dt*1e6:7.1f} us/step   {calc.md_stats}', flush=True)
