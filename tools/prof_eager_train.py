import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from bench_train import ethanol_batch
from newtonnet_amd.distributed import TrainStep
from newtonnet_amd.models import NewtonNet
args = [t.cuda() for t in ethanol_batch(32)]
torch.manual_seed(0)
model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda'); model.train()
step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), 1.0, 50.0, 1.0)
for _ in range(5): step(*args)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step(*args)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
