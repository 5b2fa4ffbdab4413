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


class GraphedTrainStep:
    """TrainStep replayed from HIP graphs for batches of fixed STRUCTURE (same z / batch / cell from step to step -- the
    MD17-style case: one molecule type, fixed batch size, shuffled conformations; trainer.py:301-313).

    The eager step is bound by the host (~600 small launches through Python autograd: 12.6 ms whatever the batch size);
    with the structure fixed nothing in it depends on the data: the neighbor list becomes a static candidate list (all pairs
    of every molecule, candidates beyond the cutoff masked to exactly zero, train_ops.forward_train) and the whole
    forward + double backward is captured once and replayed (4 ms at batch 32 on MI355X).  Two graphs: (1) zero_grad +
    forward + loss + backward, (2) gradient clipping + optimizer step; between them the data-parallel gradient
    all-reduce runs eagerly (allreduce_gradients), so the same class serves one GPU and DDP.  A new structure (e.g. the last,
    smaller batch of an epoch) re-captures; optimizers must be capture-safe (torch.optim.Adam(..., capturable=True)).
    """
    def __init__(self, model, optimizer, w_energy: float = 1.0, w_force: float = 50.0, clip_grad: float = 1.0,
                 group=None):
        self.model, self.optimizer = model, optimizer
        self.w_energy, self.w_force, self.clip_grad, self.group = w_energy, w_force, clip_grad, group
        self._st = None
        self.captures = 0

    # -- the two captured pieces -------------------------------------------------------------------------
    def _fwd_bwd(self, st):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(st['z'], st['pos'], st['cell'], st['batch'])
        sse_e = (out.energy - st['e']).pow(2).sum()
        sse_f = (out.gradient_force - st['f']).pow(2).sum()
        loss = self.w_energy * sse_e / st['n_e'] + self.w_force * sse_f / st['n_f']
        loss.backward()
        return loss.detach()

    def _update(self):
        if self.clip_grad:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad)
        self.optimizer.step()

    def _same_structure(self, z, cell, batch) -> bool:
        st = self._st
        if st is None or st['z'].shape != z.shape or st['cell'].shape != cell.shape:
            return False
        return bool(((st['z'] == z).all() & (st['batch'] == batch).all() & (st['cell'] == cell).all()).item())

    def _capture(self, z, pos, cell, batch, energy_label, force_label):
        from newtonnet_amd import hip
        dev = pos.device
        emb = self.model.embedding_layers.edge_embedding
        st = dict(z=z.clone(), cell=cell.clone(), batch=batch.clone(), pos=pos.detach().clone().requires_grad_(True),
                  e=energy_label.detach().clone(), f=force_label.detach().clone())
        st['n_e'], st['n_f'] = allreduce_counts(energy_label.numel(), force_label.numel(), dev, self.group)
        # static candidate list: every ordered pair of every molecule (minimum image when periodic)
        st['graph'] = hip.build_graph(st['pos'].detach(), st['cell'], st['batch'], 1.0e6, emb.embedding.frequencies,
                                      want_rbf=True)
        self._st = st
        self.model._static_train_graph = st['graph']
        try:
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):            # warm-up on a side stream (PyTorch's whole-network capture recipe);
                for _ in range(2):                   # the optimizer must create its state HERE, not inside the capture
                    self._fwd_bwd(st)                # (its zero-initialisation would be replayed every step)
                    self._update()
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            st['g1'] = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            st['pos'].grad = None
            with torch.cuda.graph(st['g1']):
                st['loss'] = self._fwd_bwd(st)
            # the update graph shares g1's memory pool: it reads the .grad tensors g1 produces
            st['g2'] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st['g2'], pool=st['g1'].pool()):
                self._update()
        finally:
            self.model._static_train_graph = None
        self.captures += 1

    def __call__(self, z, pos, cell, batch, energy_label, force_label):
        if not self.model.training:
            raise RuntimeError('GraphedTrainStep needs model.train()')
        if not self._same_structure(z, cell, batch):
            # The warm-up / capture passes run optimizer steps of their own on this batch: snapshot and restore IN PLACE (the
            # graphs hold the addresses of the parameters and of the optimizer's state tensors).
            params = [p.detach().clone() for p in self.model.parameters()]
            saved = {id(p): {k: v.detach().clone() for k, v in self.optimizer.state.get(p, {}).items() if torch.is_tensor(v)}
                     for p in self.model.parameters()}
            self._capture(z, pos, cell, batch, energy_label, force_label)
            with torch.no_grad():
                for p, q in zip(self.model.parameters(), params):
                    p.copy_(q)
                    for k, v in self.optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            old = saved[id(p)].get(k)
                            v.copy_(old) if old is not None else v.zero_()   # state created by the warm-up: back to its start
        st = self._st
        st['pos'].data.copy_(pos.detach())
        st['e'].copy_(energy_label.detach())
        st['f'].copy_(force_label.detach())
        st['g1'].replay()
        if dist.is_available() and dist.is_initialized():
            allreduce_gradients(self.model.parameters(), self.group)
        st['g2'].replay()
        return st['loss'].clone()
