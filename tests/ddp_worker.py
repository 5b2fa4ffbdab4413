"""Worker of tests/test_hip_train.py::test_two_rank_training_* (launched by torch.distributed.run, one process per rank; the
ranks SHARE cuda:0 -- the test box has one GPU).  Runs K fully fused training steps on this rank's molecule shard of a mixed
batch and writes the final flat parameters + losses to <out>/rank<r>.pt."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def global_batch():
    from bench import synthetic_md17_mixed
    return synthetic_md17_mixed(12, 7, 'cpu')


def run(z, pos, cell, batch, e_lab, f_lab, steps, group_ok, mode='graph'):
    from newtonnet_amd.distributed import FusedClipAdam, GraphedTrainStep, TrainStep
    from newtonnet_amd.models import NewtonNet
    torch.manual_seed(0)
    model = NewtonNet(output_properties=['energy', 'gradient_force']).cuda()
    model.train()
    opt = FusedClipAdam(model, lr=1e-3, max_norm=1.0)
    step = GraphedTrainStep(model, opt, 1.0, 50.0) if mode == 'graph' else TrainStep(model, opt, 1.0, 50.0)   # captured / eager
    args = [t.cuda() for t in (z, pos, cell, batch, e_lab, f_lab)]
    losses = [float(step(*args)) for _ in range(steps)]
    torch.cuda.synchronize()
    return model._flat_params.detach().cpu(), losses, float(opt.dev_state[1])


def main():
    out, backend, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else 'graph'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('gloo')
    from newtonnet_amd.distributed import shard_molecules
    z, pos, cell, batch, e_lab, f_lab = global_batch()
    sizes = torch.bincount(batch).tolist()
    m0, m1 = shard_molecules(sizes, world)[rank]
    a0, a1 = sum(sizes[:m0]), sum(sizes[:m1])
    flat, losses, gnorm = run(z[a0:a1], pos[a0:a1], cell[m0:m1], batch[a0:a1] - m0, e_lab[m0:m1], f_lab[a0:a1], steps, True, mode)
    torch.save(dict(flat=flat, losses=losses, gnorm=gnorm, shard=(m0, m1)), os.path.join(out, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
