"""Tooling: us per step of back-to-back eval calls on ~21.5k atoms split into molecules of n atoms (chains 1.1 A apart), for the
crossover of the molecule-resident edge kernels in molecule SIZE (run once with NNHIP_FORCE_FWD_MOL=0 NNHIP_MSG_BWD_MOL=0, once without).
usage (through gpurun): python tools/sweep_mol_size.py [n,n,...]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
gen = torch.Generator().manual_seed(1)
for n in [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else '3,6,9,12,16,21'.split(','))]:
    B = 21504 // n
    m = int(round(n ** (1.0 / 3.0))) + 1
    grid = torch.stack(torch.meshgrid(*[torch.arange(m)] * 3, indexing='ij'), dim=-1).reshape(-1, 3)[:n].float()
    pos = (1.5 * grid).repeat(B, 1) + 0.1 * torch.randn(B * n, 3, generator=gen) + 40.0 * torch.repeat_interleave(torch.rand(B, 3, generator=gen) * 50, n, dim=0)
    z = torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (B * n,), generator=gen)]
    batch = torch.repeat_interleave(torch.arange(B), n)
    args = (z.cuda(), pos.cuda(), torch.zeros(B, 3, 3, device='cuda'), batch.cuda())
    for _ in range(20): model(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter(); k = 100
    for _ in range(k): o = model(*args)
    torch.cuda.synchronize()
    print(f'n={n:3d} B={B:5d} N={B*n:6d} E={o.edge_index.shape[1]:7d}: {(time.perf_counter()-t0)/k*1e6:8.1f} us/step', flush=True)
