#!/usr/bin/env python3
"""Train-step timing (BASELINE configs[2]: MD17 ethanol-shaped, batch 32 / 10; and the config-2 batch, 1024 aspirin):
loss = MSE(E) + 50 MSE(F), Adam, clip 1.0 (scripts/config.yml:45-54).  Fused path (csrc/train.hip) eager and replayed from HIP
graphs, the inference step on the same batch, and the
CPU oracle's double backward.  usage: tools/bench_train.py [--no-cpu] [--no-torch-path]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from newtonnet_amd.distributed import FusedClipAdam, GraphedTrainStep, TrainStep  # noqa: E402
from newtonnet_amd.models import NewtonNet  # noqa: E402


def ethanol_batch(B, seed=0):
    eth0 = torch.tensor([[0.00, 0.00, 0.00], [1.52, 0.00, 0.00], [2.05, 1.32, 0.00], [-0.39, 1.02, 0.00],
                         [-0.39, -0.51, 0.89], [-0.39, -0.51, -0.89], [1.90, -0.53, 0.88], [1.90, -0.53, -0.88],
                         [3.01, 1.30, 0.00]])
    g = torch.Generator().manual_seed(seed)
    pos = eth0.repeat(B, 1) + 0.1 * torch.randn(9 * B, 3, generator=g)
    z = torch.tensor([6, 6, 8, 1, 1, 1, 1, 1, 1]).repeat(B)
    batch = torch.repeat_interleave(torch.arange(B), 9)
    return z, pos, torch.zeros(B, 3, 3), batch, torch.randn(B, generator=g), torch.randn(9 * B, 3, generator=g)


def aspirin_batch(B, seed=0):
    with np.load(os.path.join(ROOT, 'tests', 'golden', 'aspirin_frames.npz')) as f:
        z0, p0 = f['z'], f['test0_pos']
    g = torch.Generator().manual_seed(seed)
    n = len(z0)
    pos = torch.from_numpy(p0).float().repeat(B, 1) + 0.05 * torch.randn(B * n, 3, generator=g)
    z = torch.from_numpy(z0).long().repeat(B)
    batch = torch.repeat_interleave(torch.arange(B), n)
    return z, pos, torch.zeros(B, 3, 3), batch, torch.randn(B, generator=g), torch.randn(n * B, 3, generator=g)


def timeit(fn, warm, reps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    no_cpu = '--no-cpu' in sys.argv
    no_torch = '--no-torch-path' in sys.argv
    for name, maker, B in (('ethanol', ethanol_batch, 10), ('ethanol', ethanol_batch, 32), ('ethanol', ethanol_batch, 256),
                           ('aspirin', aspirin_batch, 128), ('aspirin', aspirin_batch, 1024)):
        data = maker(B)
        args = [t.cuda() for t in data]
        res = {}
        for path in ('fused',):      # (the torch-graph path of round 1 was removed in round 6)
            torch.manual_seed(0)
            model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
            model.train()
            step = TrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), 1.0, 50.0, 1.0)
            res[path + '_eager'] = timeit(lambda: step(*args), 3, 10)
            torch.manual_seed(0)
            mg = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
            mg.train()
            gstep = GraphedTrainStep(mg, torch.optim.Adam(mg.parameters(), lr=1e-3, capturable=True), 1.0, 50.0, 1.0)
            res[path + '_graph'] = timeit(lambda: gstep(*args), 3, 30)
        torch.manual_seed(0)
        mf = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
        mf.train()
        fstep = GraphedTrainStep(mf, FusedClipAdam(mf, lr=1e-3, max_norm=1.0), 1.0, 50.0, assume_static=True)
        res['all_hip_graph'] = timeit(lambda: fstep(*args), 3, 30)
        torch.manual_seed(0)
        mn = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
        mn.train()
        nstep = TrainStep(mn, FusedClipAdam(mn, lr=1e-3, max_norm=1.0), 1.0, 50.0)
        res['all_hip_eager'] = timeit(lambda: nstep(*args), 3, 30)
        os.environ['NNHIP_TRAIN_BF16'] = '1'           # bf16-operand weight-gradient products (fp32 everywhere else)
        torch.manual_seed(0)
        mb = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
        mb.train()
        bstep = TrainStep(mb, torch.optim.Adam(mb.parameters(), lr=1e-3), 1.0, 50.0, 1.0)
        res['fused_eager_bf16wgrad'] = timeit(lambda: bstep(*args), 3, 10)
        os.environ.pop('NNHIP_TRAIN_BF16')
        model.eval()
        with torch.no_grad():
            res['inference'] = timeit(lambda: model(*args[:4]), 3, 20)
        line = f'{name} B={B:5d} ({args[1].shape[0]:6d} atoms): ' + ' | '.join(f'{k} {v * 1e3:8.3f} ms' for k, v in res.items())
        line += f' | train/inference {res["all_hip_graph"] / res["inference"]:.2f}x'
        if not no_cpu and args[1].shape[0] <= 3000:
            from oracle import newtonnet_ref as ref
            torch.set_num_threads(min(16, os.cpu_count()))
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            ref.training_loss_grads(sd, *data)
            t0 = time.perf_counter()
            for _ in range(3):
                ref.training_loss_grads(sd, *data)
            cpu = (time.perf_counter() - t0) / 3
            line += f' | CPU oracle loss+grads {cpu * 1e3:8.1f} ms ({torch.get_num_threads()} threads)'
        print(line, flush=True)


if __name__ == '__main__':
    main()
