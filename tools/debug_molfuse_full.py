#!/usr/bin/env python3
"""Full-size (1024 aspirin conformers, ckpt weights) check of the row path and the fused path against the fp64 oracle, conformer by
conformer.  usage: python tools/debug_molfuse_full.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tests.test_hip_parity import make_model
from oracle import newtonnet_ref as ref
a = util.load_npz('aspirin_frames.npz')
B, n = 1024, 21
gen = torch.Generator().manual_seed(0)
pos = torch.from_numpy(a['test0_pos']).float().repeat(B, 1) + 0.05 * torch.randn(B * n, 3, generator=gen)
z = torch.from_numpy(a['z']).long().repeat(B)
batch = torch.repeat_interleave(torch.arange(B), n)
model, sd = make_model('ckpt')
args = (z.cuda(), pos.cuda(), torch.zeros(B, 3, 3, device='cuda'), batch.cuda())
torch.set_num_threads(16)
o = ref.energy_forces({k: v.double() for k, v in sd.items()}, z, pos.double(), torch.zeros(B, 3, 3, dtype=torch.float64), batch)
fr = o['forces'].view(B, n, 3)
pairs = np.bincount(batch.numpy()[o['edge_index'][0].numpy()], minlength=B) // 2
res = {}
for mode in (0, 1, 2, 3):
    os.environ['NNHIP_MOL_FUSED'] = str(mode)
    for rep in range(2):
        out = model(*args)
        f = out.gradient_force.cpu().double().view(B, n, 3)
        e = out.energy.cpu().double()
        d = (f - fr).abs().amax(dim=(1, 2))
        worst = torch.argsort(d, descending=True)[:5]
        print(f'mode {mode} rep {rep}: max |dF| vs fp64 per conformer: max {d.max():.2e} mean {d.mean():.2e}; worst', [(int(k), f'{d[k]:.1e}', int(pairs[k])) for k in worst],
              f'dE max {(e - o["energy"]).abs().max():.2e}', flush=True)
    res[mode] = f
for mode in (1, 2, 3):
    d = (res[mode] - res[0]).abs().amax(dim=(1, 2))
    worst = torch.argsort(d, descending=True)[:8]
    print(f'mode {mode} vs mode 0: worst conformers', [(int(k), f'{d[k]:.1e}', int(pairs[k])) for k in worst])
print('pairs histogram', np.bincount(pairs)[130:170])
