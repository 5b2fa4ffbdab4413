"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: flat gradient all-reduce with global loss
normalisation reproduces the single-process gradient; molecule sharding is a balanced partition."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from newtonnet_amd.distributed import TrainStep, allreduce_counts, allreduce_gradients, shard_molecules


class ToyModel(torch.nn.Module):
    """Stands in for NewtonNet on CPU: energy = sum over atoms of an MLP of pos, force = -dE/dpos (create_graph)."""
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = torch.nn.Sequential(torch.nn.Linear(3, 8), torch.nn.SiLU(), torch.nn.Linear(8, 1))

    def forward(self, z, pos, cell, batch):
        e_atom = self.net(pos).reshape(-1)
        energy = torch.zeros(cell.shape[0]).index_add_(0, batch, e_atom)
        (g,) = torch.autograd.grad(energy.sum(), pos, create_graph=True)
        return type('Out', (), dict(energy=energy, gradient_force=-g))()


def make_data():
    g = torch.Generator().manual_seed(1)
    sizes = [3, 5, 2, 7]                       # mixed molecule sizes: ranks hold different element counts
    pos = torch.randn(sum(sizes), 3, generator=g)
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    e_lab = torch.randn(len(sizes), generator=g)
    f_lab = torch.randn(sum(sizes), 3, generator=g)
    return sizes, pos, batch, e_lab, f_lab


def single_process_grads():
    sizes, pos, batch, e_lab, f_lab = make_data()
    model = ToyModel()
    p = pos.clone().requires_grad_(True)
    out = model(None, p, torch.zeros(len(sizes), 3, 3), batch)
    loss = torch.nn.functional.mse_loss(out.energy, e_lab) + 50.0 * torch.nn.functional.mse_loss(out.gradient_force, f_lab)
    loss.backward()
    return [q.grad.clone() for q in model.parameters()]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sizes, pos, batch, e_lab, f_lab = make_data()
    (m0, m1) = [(0, 2), (2, 4)][rank]
    a0, a1 = sum(sizes[:m0]), sum(sizes[:m1])
    model = ToyModel()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)      # lr 0: keep the all-reduced grads for inspection
    step = TrainStep(model, opt, w_energy=1.0, w_force=50.0, clip_grad=0.0)
    step(None, pos[a0:a1], torch.zeros(m1 - m0, 3, 3), batch[a0:a1] - m0, e_lab[m0:m1], f_lab[a0:a1])
    q.put((rank, [p.grad.numpy().copy() for p in model.parameters()]))   # numpy: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get() for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = single_process_grads()
    for r in range(2):
        for a, b in zip(got[r], want):
            assert torch.allclose(torch.from_numpy(a), b, rtol=1e-5, atol=1e-6), (torch.from_numpy(a) - b).abs().max()
    for a, b in zip(got[0], got[1]):
        assert (a == b).all()                    # replicas hold bit-identical gradients after the all-reduce


def test_shard_molecules_partition():
    sizes = [21] * 10 + [9] * 30 + [12] * 7
    for world in (1, 2, 3, 8):
        b = shard_molecules(sizes, world)
        assert b[0][0] == 0 and b[-1][1] == len(sizes)
        assert all(b[k][1] == b[k + 1][0] for k in range(world - 1))
        work = [sum(n * n for n in sizes[s:e]) for s, e in b]
        if world > 1:
            assert max(work) <= 1.5 * sum(work) / world + 21 * 21


def test_no_process_group_is_a_noop():
    assert allreduce_counts(3, 9, 'cpu') == (3.0, 9.0)
    lin = torch.nn.Linear(2, 2)
    lin.weight.grad = torch.ones_like(lin.weight)
    allreduce_gradients(lin.parameters())
    assert torch.all(lin.weight.grad == 1) and torch.all(lin.bias.grad == 0)
