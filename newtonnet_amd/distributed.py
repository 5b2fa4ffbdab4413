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
    """Global number of energy / force-component elements in this step as host floats (one tiny all-reduce + a host sync: the
    torch-autograd step uses it; the all-HIP steps keep the counts on the device, `_CountReduce`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(n_energy), float(n_force)            # (no device work, no sync)
    t = torch.tensor([float(n_energy), float(n_force)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t[0]), float(t[1])


class _CountReduce:
    """(w_E / n_E, w_F / n_F) with GLOBAL element counts as a DEVICE tensor, without a host sync: the two local counts go to the
    device, their all-reduce is issued asynchronously at the top of the step (it runs on the collective's own stream while the
    value sweeps execute) and `norm()` joins it right before the loss kernel, the first consumer.  Off the critical path, exact
    for batches whose sizes differ from rank to rank and from step to step (mixed MD17)."""

    def __init__(self, w_energy: float, w_force: float, group=None):
        self.w, self.group = (float(w_energy), float(w_force)), group
        self._w_dev = None

    def start(self, n_energy: int, n_force: int, device):
        self._local = (max(int(n_energy), 1), max(int(n_force), 1))
        self._work = self._counts = None
        if dist.is_available() and dist.is_initialized():
            self._counts = torch.tensor([float(n_energy), float(n_force)], dtype=torch.float64).to(device, non_blocking=True)
            self._work = dist.all_reduce(self._counts, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._device = device

    def norm(self) -> torch.Tensor:
        if self._counts is None:
            return torch.tensor([self.w[0] / self._local[0], self.w[1] / self._local[1]],
                                dtype=torch.float32).to(self._device, non_blocking=True)
        self._work.wait()                       # (stream-side join for RCCL; gloo completes on the host)
        if self._w_dev is None or self._w_dev.device != self._counts.device:
            self._w_dev = torch.tensor(self.w, dtype=torch.float64, device=self._counts.device)
        return (self._w_dev / self._counts.clamp_min(1.0)).float()


def _loss_spec(loss_modes, huber_delta):
    """('mse' | 'mae' | 'huber') for the energy and the force term (newtonnet/train/loss.py:53-103) -> ids of the C ABI."""
    from newtonnet_amd import hip
    modes = (loss_modes, loss_modes) if isinstance(loss_modes, str) else tuple(loss_modes)
    deltas = (huber_delta, huber_delta) if isinstance(huber_delta, (int, float)) else tuple(huber_delta)
    for m in modes:
        if m not in hip.LOSS_MODES:
            raise ValueError(f'loss mode {m} not implemented')          # (the reference's message, loss.py:77)
    return modes, (hip.LOSS_MODES[modes[0]], hip.LOSS_MODES[modes[1]]), (float(deltas[0]), float(deltas[1]))


def _torch_loss_sum(pred, label, mode, delta):
    """sum over elements of the reference's elementwise loss (the caller divides by the global element count)"""
    d = pred - label
    if mode == 'mse':
        return d.pow(2).sum()
    if mode == 'mae':
        return d.abs().sum()
    return torch.nn.functional.huber_loss(pred, label, reduction='sum', delta=delta)


def _flat_view_of(params) -> Optional[torch.Tensor]:
    """The single contiguous fp32 tensor the gradients are views of, when they are laid out back to back in parameter order
    (the fused training path, train_fused.py, produces them that way); else None."""
    if any(p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous() for p in params):
        return None
    base = params[0].grad
    st, ptr = base.untyped_storage(), base.data_ptr()
    for p in params:
        if p.grad.untyped_storage().data_ptr() != st.data_ptr() or p.grad.data_ptr() != ptr:
            return None
        ptr += 4 * p.grad.numel()
    n = (ptr - base.data_ptr()) // 4
    return torch.as_strided(base, (n,), (1,))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None) -> Optional[torch.Tensor]:
    """One flat all-reduce(sum) over all parameter gradients (in place).  Returns the flat buffer (None when there is no
    process group: nothing to do).  Parameters without a gradient get zeros, with or without a process group."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return None
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    if not (dist.is_available() and dist.is_initialized()):
        return None
    flat = _flat_view_of(params)
    if flat is not None:                      # gradients already live in one buffer: reduce it where it is
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return flat
    flat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return flat


_UNSUPPORTED_FUSED = ("FusedClipAdam / the fused step serves output_properties within {'energy', 'gradient_force', 'direct_force'} "
                      "(with 'energy'), layer_norm on every interaction layer or on none")


def _force_key(model):
    """The force the step's loss is on: 'gradient_force' when the model has that head, else 'direct_force', else None (an
    energy-only model, loss.py:30-47).  A model with BOTH force heads is refused: its two loss terms go through the model's
    train-mode forward and a torch loss -- the same kernels behind one autograd node."""
    keys = list(getattr(model, 'output_properties', ('energy', 'gradient_force')))
    if 'gradient_force' in keys and 'direct_force' in keys:
        # The reference's loss factory sums a term per configured head (loss.py:30-47).  The fused step carries ONE force term:
        # with both heads it would train the direct-force head with a zero gradient -- refuse rather than do that silently.
        raise NotImplementedError(
            "the fused training step carries one force term; a model with both 'gradient_force' and 'direct_force' trains "
            "through its train-mode forward (one autograd node on the same kernels) and a torch loss over both heads")
    return 'gradient_force' if 'gradient_force' in keys else ('direct_force' if 'direct_force' in keys else None)


class TrainStep:
    """One optimisation step with the reference's loss (scripts/config.yml:45-51; trainer.py:301-313; loss.py:5-103):
    loss = w_E * l_E(E) + w_F * l_F(F) with l = MSE / MAE / Huber (mean reduction), clip_grad_norm_, optimizer.step --
    data-parallel over molecules."""
    def __init__(self, model, optimizer, w_energy: float = 1.0, w_force: float = 50.0, clip_grad: float = 1.0,
                 group=None, loss_modes='mse', huber_delta=1.0):
        """optimizer: a torch optimizer (the step is then autograd through the model's fused node + torch clipping + the
        optimizer), or a FusedClipAdam (its max_norm is the clipping; `clip_grad` is ignored): the step then runs with NO
        autograd and no per-parameter tensors -- exact neighbor list, value sweeps, loss and its gradient, tangent sweeps,
        weight gradients, one all-reduce of the flat gradient, clip + Adam -- for batches of any (changing) structure.
        loss_modes: 'mse' | 'mae' | 'huber', or a pair (energy term, force term); huber_delta likewise."""
        self.model, self.optimizer = model, optimizer
        self.w_energy, self.w_force, self.clip_grad, self.group = w_energy, w_force, clip_grad, group
        self.loss_modes, self._mode_ids, self.huber_delta = _loss_spec(loss_modes, huber_delta)
        self.fused = isinstance(optimizer, FusedClipAdam)
        self._loss = self._gE = self._gF = None
        self._counts = _CountReduce(w_energy, w_force, group)

    def _call_fused(self, z, pos, cell, batch, energy_label, force_label):
        from newtonnet_amd import hip, train_fused
        model, dev = self.model, pos.device
        if not model.training:
            raise RuntimeError('TrainStep needs model.train()')
        if not train_fused.supported(model, list(model.output_properties)):
            raise NotImplementedError(_UNSUPPORTED_FUSED)
        emb = model.embedding_layers.edge_embedding
        force_key = _force_key(model)
        # the step's two collectives: the 2-scalar count all-reduce starts NOW and overlaps the value sweeps (joined before the
        # loss kernel, no host sync); the flat gradient all-reduce is the one on the critical path
        self._counts.start(energy_label.numel(), force_label.numel() if force_key else 0, dev)
        zc = z.contiguous() if z.dtype == torch.int64 else z.long().contiguous()
        bc = batch.contiguous() if batch.dtype == torch.int64 else batch.long().contiguous()
        pd, cd = hip._f32c(pos.detach(), 'pos'), hip._f32c(cell.detach(), 'cell')
        N, B = pd.shape[0], cd.shape[0]
        e_lab = hip._f32c(energy_label.detach(), 'energy_label')
        f_lab = hip._f32c(force_label.detach(), 'force_label') if force_key else None
        if e_lab.numel() != B or (f_lab is not None and f_lab.numel() != 3 * N):
            raise ValueError(f'labels of shape {tuple(energy_label.shape)}, {tuple(getattr(force_label, "shape", ()))} for {B} molecules, '
                             f'{N} atoms')
        with torch.no_grad():
            g = hip.build_graph(pd, cd, bc, emb.cutoff, emb.embedding.frequencies, want_rbf=True, z=zc, envelope=emb.envelope_id)
            ws = train_fused.acquire_workspace(model, g, dev)
            if self._loss is None or self._loss.device != dev:
                self._loss = torch.zeros(1, dtype=torch.float32, device=dev)
            if self._gE is None or self._gE.shape[0] != B or self._gF.shape[0] != N or self._gE.device != dev:
                self._gE = torch.empty(B, dtype=torch.float32, device=dev)
                self._gF = torch.empty(N, 3, dtype=torch.float32, device=dev)
            runner = train_fused.Runner(model, zc, pd, cd, bc, g, ws)
            runner.values()
            norm = self._counts.norm()
            pred_f = ws.dfh_out if force_key == 'direct_force' else ws.forces      # the force the loss is on (loss.py:36-47)
            hip._check(hip.lib().nnhip_loss_grad(hip._ptr(ws.energy), hip._ptr(e_lab), B, hip._ptr(pred_f),
                                                 hip._ptr(f_lab if f_lab is not None else pred_f), 3 * N if force_key else 0,
                                                 hip._ptr(norm), self._mode_ids[0], self._mode_ids[1],
                                                 self.huber_delta[0], self.huber_delta[1], hip._ptr(self._loss),
                                                 hip._ptr(self._gE), hip._ptr(self._gF), hip._stream(dev)), 'nnhip_loss_grad')
            runner.grads(self._gE, self._gF if force_key == 'gradient_force' else None,
                         self._gF if force_key == 'direct_force' else None)
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(ws.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.optimizer.step(ws.flat_grad)
            return self._loss[0].clone()

    def __call__(self, z, pos, cell, batch, energy_label, force_label):
        if self.fused:
            return self._call_fused(z, pos, cell, batch, energy_label, force_label)
        self.optimizer.zero_grad(set_to_none=True)
        n_e, n_f = allreduce_counts(energy_label.numel(), force_label.numel() if force_label is not None else 0, pos.device,
                                    self.group)
        pos = pos.detach().clone().requires_grad_(True)
        out = self.model(z, pos, cell, batch)
        force_key = _force_key(self.model)
        loss = self.w_energy * _torch_loss_sum(out.energy, energy_label, self.loss_modes[0], self.huber_delta[0]) / n_e
        if force_key:                                                        # this rank's share of the global loss
            loss = loss + self.w_force * _torch_loss_sum(getattr(out, force_key), force_label, self.loss_modes[1],
                                                         self.huber_delta[1]) / n_f
        loss.backward()
        allreduce_gradients(self.model.parameters(), self.group)
        if self.clip_grad:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], self.clip_grad)
        self.optimizer.step()
        return loss.detach()


def flatten_parameters(model) -> torch.Tensor:
    """Re-home the trainable parameters as views of ONE flat fp32 tensor, in parameter order (the layout of the flat gradient the
    fused training path produces): clip + Adam then run as two launches over the flat buffers and the data-parallel all-reduce
    moves one tensor.  Idempotent; `model.to(...)` afterwards breaks the aliasing and the next call restores it."""
    from newtonnet_amd.train_fused import trainable_parameters
    params = trainable_parameters(model)
    flat = getattr(model, '_flat_params', None)
    if flat is not None and flat.device == params[0].device:
        ptr, ok = flat.data_ptr(), True
        for p in params:
            ok = ok and p.data_ptr() == ptr and p.is_contiguous()
            ptr += 4 * p.numel()
        if ok:
            return flat
    flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in params])
    off = 0
    for p in params:
        p.data = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    model.__dict__['_flat_params'] = flat
    return flat


class FusedClipAdam(torch.optim.Optimizer):
    """clip_grad_norm_(max_norm) + Adam (torch.optim.Adam defaults: no weight decay, no amsgrad; trainer.py:311-313) as two HIP
    launches over the model's flat parameter buffer (csrc/train.hip: gradnorm_partial_kernel, clip_adam_kernel).

    A torch.optim.Optimizer: `param_groups[0]['lr']` is the learning rate the next step uses, so the reference's schedulers
    (ReduceLROnPlateau etc., trainer.py:190,254) and its `lr <= min_lr` stop criterion drive it unchanged.  The step counter AND
    the hyper-parameters (lr, betas, eps, max_norm) live in device memory, so an update captured into a HIP graph follows the
    schedule: `sync()` refreshes them (a 20-byte copy when something changed) and GraphedTrainStep calls it before every replay.
    Parameters with requires_grad False (the freeze_* switches of scripts/newtonnet_train.py:69-81) are masked on the device:
    they do not enter the clipping norm and are not updated, like parameters a torch optimizer was never given.
    `step(flat_grad)` takes the flat gradient of the fused training path; `step()` gathers the parameters' .grad tensors."""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0):
        from newtonnet_amd import hip
        from newtonnet_amd.train_fused import trainable_parameters
        self.model = model
        self.flat = flatten_parameters(model)
        self._params = trainable_parameters(model)
        super().__init__(self._params, dict(lr=float(lr), betas=tuple(betas), eps=float(eps), max_norm=float(max_norm or 0.0)))
        dev = self.flat.device
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.dev_state = torch.zeros(2, dtype=torch.float32, device=dev)      # (step, last gradient norm)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)          # (lr, beta1, beta2, eps, max_norm)
        self.mask = torch.ones(self.flat.numel(), dtype=torch.uint8, device=dev)
        self.scratch = torch.empty(hip.lib().nnhip_clip_adam_scratch_bytes() // 4, dtype=torch.float32, device=dev)
        self._hyper_host = self._mask_key = None
        self.sync()

    # plain attributes of the round-2 API, now views of the param group
    lr = property(lambda self: self.param_groups[0]['lr'], lambda self, v: self.param_groups[0].__setitem__('lr', float(v)))
    betas = property(lambda self: self.param_groups[0]['betas'])
    eps = property(lambda self: self.param_groups[0]['eps'])
    max_norm = property(lambda self: self.param_groups[0]['max_norm'],
                        lambda self, v: self.param_groups[0].__setitem__('max_norm', float(v or 0.0)))

    def sync(self):
        """Bring the device-side hyper-parameters and the frozen-parameter mask up to date with param_groups / requires_grad
        (host compare; device copies only on change).  Not capturable: call it outside a graph capture."""
        g = self.param_groups[0]
        h = (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['max_norm'] or 0.0))
        if not (h[0] >= 0.0 and 0.0 <= h[1] < 1.0 and 0.0 <= h[2] < 1.0):
            raise ValueError(f'FusedClipAdam: invalid hyper-parameters lr={h[0]}, betas=({h[1]}, {h[2]})')
        if h != self._hyper_host:
            self.hyper.copy_(torch.tensor(h + (0.0, 0.0, 0.0), dtype=torch.float32), non_blocking=False)
            self._hyper_host = h
        key = tuple(p.requires_grad for p in self._params)
        if key != self._mask_key:
            m = torch.cat([torch.full((p.numel(),), 1 if p.requires_grad else 0, dtype=torch.uint8) for p in self._params])
            self.mask.copy_(m)
            self._mask_key = key
            self._masked = not all(key)

    def _gather_grads(self) -> torch.Tensor:
        flat = _flat_view_of([p for p in self._params]) if all(p.grad is not None for p in self._params) else None
        if flat is not None:
            return flat
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32)
                          for p in self._params])

    def step(self, flat_grad: Optional[torch.Tensor] = None, closure=None):
        from newtonnet_amd import hip
        if closure is not None:
            raise NotImplementedError('FusedClipAdam.step does not take a closure')
        if flat_grad is None:
            flat_grad = self._gather_grads()
        if not torch.cuda.is_current_stream_capturing():
            self.flat = flatten_parameters(self.model)
            self.sync()
        if flat_grad.numel() != self.flat.numel() or flat_grad.dtype != torch.float32 or not flat_grad.is_contiguous():
            raise ValueError('FusedClipAdam.step needs the flat fp32 gradient of the model (train_fused.TrainWorkspace.flat_grad)')
        hip._check(hip.lib().nnhip_clip_adam_dev(hip._ptr(self.flat), hip._ptr(flat_grad), hip._ptr(self.exp_avg),
                                                 hip._ptr(self.exp_avg_sq), self.flat.numel(), hip._ptr(self.scratch),
                                                 hip._ptr(self.dev_state), hip._ptr(self.hyper), hip._ptr(self.mask),
                                                 hip._stream(self.flat.device)), 'nnhip_clip_adam_dev')

    def state_dict(self):
        g = self.param_groups[0]
        return dict(exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(), state=self.dev_state.clone(), lr=g['lr'],
                    betas=g['betas'], eps=g['eps'], max_norm=g['max_norm'])

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.dev_state.copy_(sd['state'])
        g = self.param_groups[0]
        g['lr'], g['betas'], g['eps'], g['max_norm'] = sd['lr'], tuple(sd['betas']), sd['eps'], sd['max_norm']
        self.sync()


class GraphedTrainStep:
    """TrainStep replayed from HIP graphs for batches of fixed STRUCTURE (same z / batch / cell from step to step -- the
    MD17-style case: one molecule type, fixed batch size, shuffled conformations; trainer.py:301-313).

    The eager step is bound by the host (~600 small launches through Python autograd: 12.6 ms whatever the batch size);
    with the structure fixed nothing in it depends on the data: the neighbor list becomes a static candidate list (all pairs
    of every molecule, candidates beyond the cutoff masked to exactly zero, train_fused.forward_train) and the whole
    forward + double backward is captured once and replayed (4 ms at batch 32 on MI355X).  Two graphs: (1) zero_grad +
    forward + loss + backward, (2) gradient clipping + optimizer step; between them the data-parallel gradient
    all-reduce runs eagerly (allreduce_gradients), so the same class serves one GPU and DDP.  A new structure (e.g. the last,
    smaller batch of an epoch) re-captures; optimizers must be capture-safe (torch.optim.Adam(..., capturable=True)).
    """
    EAGER_ABOVE_ATOMS = 8192     # fused mode: batches at least this large run eagerly on the exact neighbor list

    def __init__(self, model, optimizer, w_energy: float = 1.0, w_force: float = 50.0, clip_grad: float = 1.0,
                 group=None, assume_static: bool = False, loss_modes='mse', huber_delta=1.0):
        """optimizer: a capturable torch optimizer (the step is then torch autograd + torch optimizer, captured), or a
        FusedClipAdam (its max_norm is the clipping; `clip_grad` is ignored): the whole step then runs on hand-written kernels
        with NO autograd -- value sweeps, the loss and its gradient, tangent sweeps, weight gradients, clip + Adam.
        assume_static: the structure (z / batch / cell, hence the global element counts) never changes after the first step on
        any rank: no per-step structure check (a device->host sync) and no per-step count all-reduce -- ONE collective per step,
        the flat gradient.  loss_modes / huber_delta: as TrainStep."""
        self.model, self.optimizer = model, optimizer
        self.w_energy, self.w_force, self.clip_grad, self.group = w_energy, w_force, clip_grad, group
        self.loss_modes, self._mode_ids, self.huber_delta = _loss_spec(loss_modes, huber_delta)
        self.assume_static = assume_static
        self.fused = isinstance(optimizer, FusedClipAdam)
        self._st = None
        self._eager = None
        self._use_eager = None
        self.captures = 0

    # -- fully fused mode ------------------------------------------------------------------------------------
    def _capture_fused(self, z, pos, cell, batch, energy_label, force_label, norm):
        from newtonnet_amd import hip, train_fused
        model = self.model
        if not train_fused.supported(model, list(model.output_properties)):
            raise NotImplementedError(_UNSUPPORTED_FUSED)
        force_key = _force_key(model)
        dev = pos.device
        emb = model.embedding_layers.edge_embedding
        flatten_parameters(model)
        st = dict(z=z.long().contiguous().clone(), cell=cell.float().contiguous().clone(),
                  batch=batch.long().contiguous().clone(), pos=pos.detach().float().contiguous().clone(),
                  e=energy_label.detach().float().contiguous().clone(),
                  f=(force_label.detach().float().contiguous().clone() if force_key else torch.zeros(1, device=dev)))
        N, B = st['pos'].shape[0], st['cell'].shape[0]
        st['graph'] = hip.build_graph(st['pos'], st['cell'], st['batch'], 1.0e6, emb.embedding.frequencies, want_rbf=True,
                                      z=st['z'], envelope=emb.envelope_id)                      # static candidate list: all pairs of every molecule
        g = st['graph']
        ws = train_fused.TrainWorkspace(model, N, g.n_edges, B, dev)
        runner = train_fused.Runner(model, st['z'], st['pos'], st['cell'], st['batch'], g, ws)
        st.update(ws=ws, runner=runner, norm=torch.zeros(2, dtype=torch.float32, device=dev),
                  loss=torch.zeros(1, dtype=torch.float32, device=dev), gE=torch.empty(B, dtype=torch.float32, device=dev),
                  gF=torch.empty(N, 3, dtype=torch.float32, device=dev))
        L_ = hip.lib()

        def body():
            hip.refresh_graph(g, st['pos'], st['cell'], st['batch'], emb.cutoff, emb.embedding.frequencies)
            runner.values()
            pred_f = ws.dfh_out if force_key == 'direct_force' else ws.forces
            hip._check(L_.nnhip_loss_grad(hip._ptr(ws.energy), hip._ptr(st['e']), B, hip._ptr(pred_f), hip._ptr(st['f']),
                                          3 * N if force_key else 0, hip._ptr(st['norm']), self._mode_ids[0], self._mode_ids[1],
                                          self.huber_delta[0], self.huber_delta[1], hip._ptr(st['loss']), hip._ptr(st['gE']),
                                          hip._ptr(st['gF']), hip._stream(dev)), 'nnhip_loss_grad')
            runner.grads(st['gE'], st['gF'] if force_key == 'gradient_force' else None,
                         st['gF'] if force_key == 'direct_force' else None)
        self._st = st
        st['norm'].copy_(norm)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            body()                                   # warm-up: one-time kernel attributes, lazy module loads
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        st['g1'] = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(st['g1'], capture_error_mode='thread_local'):
            body()
        st['g2'] = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(st['g2'], pool=st['g1'].pool(), capture_error_mode='thread_local'):
            self.optimizer.step(ws.flat_grad)
        self.captures += 1

    def _agree_eager(self, n_atoms: int, dev) -> bool:
        """Eager (exact list) or replay (static candidate list)?  Decided ONCE, on the first call, from the LARGEST per-rank atom
        count (one MAX all-reduce): every rank takes the same path -- the two paths issue different collectives, so a rank-local
        decision with uneven shards around the threshold would leave them unmatched."""
        if not (dist.is_available() and dist.is_initialized()):
            # one process: nothing to agree on -- decide per call (a small first batch must not pin later large ones, or one
            # large molecule, onto the static all-pairs candidate list, which grows with sum n_i^2)
            return n_atoms >= self.EAGER_ABOVE_ATOMS
        if self._use_eager is None:
            t = torch.tensor([float(n_atoms)], dtype=torch.float32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self._use_eager = int(t.item()) >= self.EAGER_ABOVE_ATOMS
        return self._use_eager

    def _call_fused(self, z, pos, cell, batch, energy_label, force_label):
        """Per step: (1) unless assume_static: agree on the global loss normalisation and on whether ANY rank's batch structure
        changed (then all ranks re-capture together: a rank-local decision would leave collectives unmatched) with one tiny
        all-reduce -- with assume_static the normalisation of the first step is kept and there is no such collective;
        (2) replay forward + loss + gradients; (3) all-reduce the flat gradient; (4) replay clip + Adam."""
        distributed = dist.is_available() and dist.is_initialized()
        if self._agree_eager(pos.shape[0], pos.device):
            # device-bound sizes: replay saves nothing, and the static all-pairs candidate list is larger than the exact one
            # (1024 aspirin conformers: 8.7 ms replayed vs 7.2 ms eager) -- run the same step eagerly on the exact list
            if self._eager is None:
                self._eager = TrainStep(self.model, self.optimizer, self.w_energy, self.w_force, self.clip_grad, self.group,
                                        self.loss_modes, self.huber_delta)
            return self._eager(z, pos, cell, batch, energy_label, force_label)
        if self.assume_static and self._st is not None:
            st = self._st                     # structure and global counts are those of the capture: nothing to agree on
        else:
            changed = self._st is None or not (self.assume_static or self._same_structure(z, cell, batch))
            norm, changed = self._global_norm(energy_label, force_label, changed, pos.device)
            if changed:
                self._capture_fused(z, pos, cell, batch, energy_label, force_label, norm)
            st = self._st
            st['norm'].copy_(norm)
        st['pos'].copy_(pos.detach(), non_blocking=True)
        st['e'].copy_(energy_label.detach(), non_blocking=True)
        if force_label is not None and st['f'].numel() == force_label.numel():      # (energy-only models carry no force label)
            st['f'].copy_(force_label.detach(), non_blocking=True)
        st['g1'].replay()
        if distributed:
            dist.all_reduce(st['ws'].flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        self.optimizer.sync()                 # learning-rate schedule / freeze switches reach the captured update
        st['g2'].replay()
        return st['loss'][0].clone()

    # -- the two captured pieces (torch autograd + torch optimizer) -------------------------------------------------------------------------
    def _fwd_bwd(self, st):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(st['z'], st['pos'], st['cell'], st['batch'])
        sse_e = _torch_loss_sum(out.energy, st['e'], self.loss_modes[0], self.huber_delta[0])
        force_key = _force_key(self.model)
        loss = st['norm'][0] * sse_e                               # (w_E / n_E, w_F / n_F): a DEVICE tensor, refreshed every step
        if force_key:
            loss = loss + st['norm'][1] * _torch_loss_sum(getattr(out, force_key), st['f'], self.loss_modes[1], self.huber_delta[1])
        loss.backward()
        return loss.detach()

    def _update(self):
        if self.clip_grad:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], self.clip_grad)
        self.optimizer.step()

    def _same_structure(self, z, cell, batch) -> bool:
        st = self._st
        if st is None or st['z'].shape != z.shape or st['cell'].shape != cell.shape:
            return False
        return bool(((st['z'] == z).all() & (st['batch'] == batch).all() & (st['cell'] == cell).all()).item())

    def _global_norm(self, energy_label, force_label, changed: bool, dev):
        """(w_E / n_E, w_F / n_F) with the GLOBAL element counts as a device tensor, and whether ANY rank's batch structure
        changed -- one small all-reduce per step under torch.distributed (every rank must take the same re-capture decision,
        or the collectives inside the capture's warm-up would be unmatched)."""
        n_f = force_label.numel() if (force_label is not None and _force_key(self.model)) else 0
        if not (dist.is_available() and dist.is_initialized()):       # nothing to agree on: no device work
            return torch.tensor([self.w_energy / max(energy_label.numel(), 1), self.w_force / max(n_f, 1)],
                                dtype=torch.float32), changed
        counts = torch.tensor([float(energy_label.numel()), float(max(n_f, 1) if n_f == 0 else n_f), 1.0 if changed else 0.0],
                              dtype=torch.float32, device=dev)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
        if not (self.assume_static and self._st is not None):
            changed = bool(counts[2].item() > 0)
        w = torch.tensor([self.w_energy, self.w_force], dtype=torch.float32, device=dev)
        return w / counts[:2], changed

    def _capture(self, z, pos, cell, batch, energy_label, force_label, norm):
        from newtonnet_amd import hip
        dev = pos.device
        emb = self.model.embedding_layers.edge_embedding
        st = dict(z=z.clone(), cell=cell.clone(), batch=batch.clone(), pos=pos.detach().clone().requires_grad_(True),
                  e=energy_label.detach().clone(), f=force_label.detach().clone(), norm=norm.to(dev).clone())
        # static candidate list: every ordered pair of every molecule (minimum image when periodic)
        st['graph'] = hip.build_graph(st['pos'].detach(), st['cell'], st['batch'], 1.0e6, emb.embedding.frequencies,
                                      want_rbf=True, envelope=emb.envelope_id)
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
            with torch.cuda.graph(st['g1'], capture_error_mode='thread_local'):
                st['loss'] = self._fwd_bwd(st)
            # the update graph shares g1's memory pool: it reads the .grad tensors g1 produces
            st['g2'] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st['g2'], pool=st['g1'].pool(), capture_error_mode='thread_local'):
                self._update()
        finally:
            self.model._static_train_graph = None
        self.captures += 1

    def __call__(self, z, pos, cell, batch, energy_label, force_label):
        if not self.model.training:
            raise RuntimeError('GraphedTrainStep needs model.train()')
        if self.fused:
            return self._call_fused(z, pos, cell, batch, energy_label, force_label)
        changed = self._st is None or not (self.assume_static or self._same_structure(z, cell, batch))
        norm, changed = self._global_norm(energy_label, force_label, changed, pos.device)
        if changed:
            # The warm-up / capture passes run optimizer steps of their own on this batch: snapshot and restore IN PLACE (the
            # graphs hold the addresses of the parameters and of the optimizer's state tensors).
            params = [p.detach().clone() for p in self.model.parameters()]
            saved = {id(p): {k: v.detach().clone() for k, v in self.optimizer.state.get(p, {}).items() if torch.is_tensor(v)}
                     for p in self.model.parameters()}
            self._capture(z, pos, cell, batch, energy_label, force_label, norm)
            with torch.no_grad():
                for p, q in zip(self.model.parameters(), params):
                    p.copy_(q)
                    for k, v in self.optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            old = saved[id(p)].get(k)
                            v.copy_(old) if old is not None else v.zero_()   # state created by the warm-up: back to its start
        st = self._st
        st['norm'].copy_(norm)
        st['pos'].data.copy_(pos.detach())
        st['e'].copy_(energy_label.detach())
        st['f'].copy_(force_label.detach())
        st['g1'].replay()
        if dist.is_available() and dist.is_initialized():
            allreduce_gradients(self.model.parameters(), self.group)
        st['g2'].replay()
        return st['loss'].clone()
