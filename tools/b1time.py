import os, sys, torch
sys.path.insert(0, '/root/repo')
import bench
from newtonnet_amd import hip
from newtonnet_amd.models import NewtonNet
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.eval()
for B in (512, 1024, 2048):
    args = bench.synthetic_aspirin(B, 0, 'cuda')
    for mode in os.environ.get('B1_MODES', '0,7').split(','):
        os.environ['NNHIP_MOL_FUSED'] = mode
        for _ in range(3): model(*args)
        torch.cuda.synchronize()
        hip.timers_enable(True)
        for _ in range(10): model(*args)
        torch.cuda.synchronize()
        tm = hip.timers_read(reset=True)
        hip.timers_enable(False)
        print(B, 'mode', mode, {k: (round(v[0] / 10 * 1e3, 1), v[1] // 10) for k, v in tm.items() if v[1]})
