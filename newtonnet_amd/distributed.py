"""Data-parallel plumbing for the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no working multi-GPU path.  Molecules never interact
(representations.py:74-78), so a batch shards by molecule:
  * inference: no collective at all (bench.py --gpus N);
  * training: replicas + ONE all-reduce(sum) per step of the flat fp32 gradient (401,135 elements = 1.6 MB; latency
    bound at that size, so a single bucket, no overlap machinery), preceded by a 2-scalar all-reduce of the loss element
    counts so that the MSE means are GLOBAL means (nn.MSELoss averages over the local batch, loss.py:72,96; with mixed
    molecule sizes an average of per-rank means would differ from the single-process loss).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_molecules(atoms_per_molecule: Sequence[int], world_size: int) -> List[Tuple[int, int]]:
    """Contiguous molecule ranges [start, end) per rank, balanced by the all-pairs work sum(n_b^2)."""
    w = [int(n) * int(n) for n in atoms_per_molecule]
    total = sum(w)
    bounds, acc, start = [], 0, 0
    for rank in range(world_size):
        target = total * (rank + 1) / world_size
        end = start
        while end < len(w) and (acc + w[end] <= target or end == start and rank < len(w)):
            acc += w[end]
            end += 1
        if rank == world_size - 1:
            end = len(w)
        bounds.append((start, end))
        start = end
    return bounds


def allreduce_counts(n_energy: int, n_force: int, device, group=None) -> Tuple[float, float]:
    """Global number of energy / force-component elements in this step (one tiny all-reduce)."""
    t = torch.tensor([float(n_energy), float(n_force)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0]), float(t[1])


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None) -> Optional[torch.Tensor]:
    """One flat all-reduce(sum) over all parameter gradients (in place).  Returns the flat buffer."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return None
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in params])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return flat


class TrainStep:
    """One optimisation step with the reference's loss (scripts/config.yml:45-51; trainer.py:301-313):
    loss = w_E * MSE(E) + w_F * MSE(F), clip_grad_norm_, optimizer.step -- data-parallel over molecules."""
    def __init__(self, model, optimizer, w_energy: float = 1.0, w_force: float = 50.0, clip_grad: float = 1.0,
                 group=None):
        self.model, self.optimizer = model, optimizer
        self.w_energy, self.w_force, self.clip_grad, self.group = w_energy, w_force, clip_grad, group

    def __call__(self, z, pos, cell, batch, energy_label, force_label):
        self.optimizer.zero_grad(set_to_none=True)
        n_e, n_f = allreduce_counts(energy_label.numel(), force_label.numel(), pos.device, self.group)
        pos = pos.detach().clone().requires_grad_(True)
        out = self.model(z, pos, cell, batch)
        sse_e = (out.energy - energy_label).pow(2).sum()
        sse_f = (out.gradient_force - force_label).pow(2).sum()
        loss = self.w_energy * sse_e / n_e + self.w_force * sse_f / n_f      # this rank's share of the global loss
        loss.backward()
        allreduce_gradients(self.model.parameters(), self.group)
        if self.clip_grad:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad)
        self.optimizer.step()
        return loss.detach()
