#!/usr/bin/env python3
"""Soak of the DEFAULT forms (what bench.py and every user runs): repeat one energy + force step on fixed inputs and require every
repeat to be bitwise the first -- aspirin batches of several sizes, a mixed MD17-shaped batch, a periodic box, and a training step.
Written after the intermittent error found in round 5's opt-in fused kernels (removed in round 6) (profiles/r05_mol_fused2_soak.txt): the same kind of
check for the kernels that are on by default.
usage: python tools/soak_default.py [reps_scale]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newtonnet_amd.models import NewtonNet
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
torch.manual_seed(0)
dev = 'cuda'
model = NewtonNet(output_properties=['energy', 'gradient_force']).to(dev)
model.eval()


def soak(name, inputs, reps):
    z, pos, cell, batch = inputs
    first, bad, t0 = None, 0, time.time()
    for k in range(reps):
        out = model(z, pos, cell, batch)
        e, f = out.energy, out.gradient_force
        if first is None:
            first = (e.clone(), f.clone())
        elif not (torch.equal(e, first[0]) and torch.equal(f, first[1])):
            bad += 1
            if bad <= 3:
                d = (f - first[1]).abs()
                print(f'   {name}: repeat {k} differs: max |dF| {d.max():.3e} in {int((d.amax(dim=1) > 0).sum())} atoms', flush=True)
    torch.cuda.synchronize()
    print(f'{name}: N = {z.shape[0]}, {reps} repeats, {bad} differ from the first ({time.time() - t0:.1f} s)', flush=True)


for B, reps in ((1, 4000), (48, 4000), (128, 4000), (256, 4000), (512, 4000), (640, 3000), (1024, 6000), (2048, 2000)):
    soak(f'aspirin x {B}', bench.synthetic_aspirin(B, 0, dev), int(reps * scale))
mixed = bench.synthetic_md17_mixed(288, 0, dev)
soak('mixed MD17 shapes x 288', mixed[:4], int(3000 * scale))
zb, pb, cb, bb = bench.synthetic_box(12000, 47, 0, dev)
soak('periodic box, 12 000 atoms of the config-5 lattice', (zb, pb, cb, bb), int(300 * scale))
