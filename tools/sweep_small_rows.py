import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
for B in [int(b) for b in (sys.argv[1].split(",") if len(sys.argv) > 1 else "48,96,144,192,256,384".split(","))]:
    z, pos, cell, batch = bench.synthetic_aspirin(B, 0, 'cuda')
    for _ in range(30): model(z, pos, cell, batch)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 300
    for _ in range(n): model(z, pos, cell, batch)
    torch.cuda.synchronize()
    print(f'B={B:4d} N={21*B:5d}: {(time.perf_counter()-t0)/n*1e6:7.1f} us/step', flush=True)
