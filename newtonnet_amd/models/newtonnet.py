"""NewtonNet on MI355X: same constructor / forward / state_dict as the reference model
(newtonnet/models/newtonnet.py:12-113), with the forward executed by hand-written HIP kernels.

The nn.Modules below only hold parameters under the reference's names, so state_dicts interchange:
  embedding_layers.node_embedding.weight, embedding_layers.edge_embedding.embedding.frequencies,
  interaction_layers.{l}.message_nodepart.{0,2}.{weight,bias}, .message_edgepart.weight,
  .equiv_message{1,2}.{0,2}.weight, .equiv_update.weight,
  output_layers.{k}.layers.{0,2,4}.{weight,bias}, scalers.{k}.{scale,shift}.weight
forward(z, pos, cell, batch) needs tensors on a ROCm device: there is no CPU implementation in this
package (the parity oracle lives in oracle/ and is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C

import os
import threading
import time

import torch
from torch import nn

from newtonnet_amd import hip
from newtonnet_amd.layers.activations import HIP_FUSED, get_activation_by_string
from newtonnet_amd.layers.representations import EdgeEmbedding
from newtonnet_amd.layers.scalers import get_scaler_by_string
from newtonnet_amd.models.output import (CustomOutputSet, DerivativeProperty, EnergyOutput, GradientForceOutput,
                                         StressOutput, VirialOutput, get_aggregator_by_string, get_output_by_string)


# A/B and debugging switches, read once (never per call)
_DEFERRED = os.environ.get('NNHIP_DEFERRED', '1') != '0'                 # 0: every eval-mode call waits for its edge count
_GRAPH_EARLY = os.environ.get('NNHIP_GRAPH_EARLY', '1') != '0'           # (synchronous path) fill queued ahead of the wait
_PREPARE_EVERY_CALL = os.environ.get('NNHIP_PREPARE_EVERY_CALL', '0') == '1'
# One big system (a single molecule of at least this many atoms): the step runs on the atoms in Morton order of cutoff-sized cells
# and hands every per-atom result back in the caller's order (0: off).  The permutation, its inverse and the permuted inputs come
# from the library's own kernels (hip.spatial_order: csrc/graph.hip, five launches; round 5 used torch's argsort); the neighbor list
# is mapped back on demand by nnhip_edge_index_unpermute.  At 16k atoms an input that is already spatially ordered loses 3 % and a
# randomly ordered one gains 16 %, from 30k atoms up nothing is lost (tools/box_order_ab.py, profiles/r05_box_order_ab.txt).
_SPATIAL_ORDER_MIN = int(os.environ.get('NNHIP_SPATIAL_ORDER_MIN', '16384'))


def _versions(tensors):
    """In-place modification counters of the input tensors (inference tensors keep none: they cannot be modified in place
    outside inference mode either, and inside it nothing can be told -- such inputs are taken as unchanged)."""
    out = []
    for t in tensors:
        try:
            out.append(t._version)
        except RuntimeError:
            out.append(None)
    return tuple(out)


_LOCK_CREATION = threading.Lock()


class _CallGuard:
    """One eval-mode call (or the repeat of one) at a time per module, ordered on the GPU behind the module's previous call.

    The module keeps per-module state between calls -- the workspace, the prepared block, the pinned slots of the deferred
    checks, the capacity hints -- and the reference's callers are single-threaded on one stream ("kernels launch on
    the current stream, no global state").  Two things make other callers safe: a re-entrant lock around the host side of a call
    (two Python threads sharing a module take turns), and, when the current stream is not the stream of the module's previous
    call, `current.wait_stream(previous)` before anything is queued (the shared workspace is then never written by two streams
    at once) plus `record_stream` on the cached tensors (the caching allocator must not hand their blocks out while this stream
    still reads them).  Costs one dict lookup and an uncontended lock per call on the usual one-stream path."""
    __slots__ = ('owner', 'device')

    def __init__(self, owner, device):
        self.owner, self.device = owner, device

    def __enter__(self):
        d = self.owner.__dict__
        lock = d.get('_call_lock')
        if lock is None:
            with _LOCK_CREATION:
                lock = d.setdefault('_call_lock', threading.RLock())
        lock.acquire()
        try:      # (anything below may raise -- a wrong device index, a freed storage: the lock must not stay held)
            cur = torch.cuda.current_stream(self.device)
            prev = d.get('_last_stream')
            if prev is not None and prev != cur:
                if prev.device == cur.device:
                    cur.wait_stream(prev)
                for name in ('_infer_ws',):
                    t = d.get(name)
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(cur)
                blk = d.get('_prep_block')
                if blk is not None and isinstance(blk[1], torch.Tensor):
                    blk[1].record_stream(cur)
            d['_last_stream'] = cur
        except BaseException:
            lock.release()
            raise
        return self

    def __exit__(self, *exc):
        self.owner.__dict__['_call_lock'].release()
        return False


class _Deferred:
    """One eval-mode forward call whose host-side checks are deferred (NewtonNet._forward_deferred).

    The reference's forward is synchronous (RadiusGraph.forward, representations.py:57-100, sizes its tensors from the edge
    count) and raises IndexError / ValueError on the spot.  Here the steady-state call queues everything and returns; the
    two words the host needs -- the edge count and the status bits -- are read when a result is first touched (`settle`) or
    when the module's next call starts (`NewtonNet._settle_last`):
      status bit 1 / 2   -> the ValueError (batch order) / IndexError (species) the synchronous call raises;
      count > capacity   -> the device emptied the graph (nnhip_graph_finish_dev): the call is repeated synchronously;
      status bit 4       -> a parameter changed since the prepared block was filled: refill and repeat likewise.
    A repeat reads the caller's input tensors again: if one of them was modified in place in the meantime the record raises
    instead of returning numbers for the wrong inputs."""
    QUEUED, WORDS, DONE = 0, 1, 2
    __slots__ = ('owner', 'inputs', 'zc', 'energy_idx', 'want_forces', 'want_virial', 'res', 'graph', 'tail', 'event', 'cap',
                 'versions', 'state', 'error', 'reported', 'count', 'bad', 'small_molecules', 'param_stamp')

    def __init__(self, owner, inputs, zc, energy_idx, want_forces, want_virial):
        self.owner, self.inputs, self.zc, self.energy_idx = owner, inputs, zc, energy_idx
        self.want_forces, self.want_virial = want_forces, want_virial
        self.res = self.graph = self.tail = self.event = self.versions = self.error = None
        self.cap = self.count = self.bad = 0
        self.small_molecules = True
        self.param_stamp = None
        self.state, self.reported = _Deferred.QUEUED, False

    def read_words(self):
        """Wait for the (count, status, change counter) words of this call -- the last neighbor-list kernel stores them into the
        pinned slot, this call's sequence number behind them -- and take them off the slot.  No kernel is launched, nothing is
        repeated here."""
        if self.state != _Deferred.QUEUED:
            return
        tail, seq = self.tail, self.event          # (a numpy view of the pinned slot; this call's sequence number)
        if int(tail[3]) != seq:
            # not there yet (the host is ahead of the GPU): poll -- the word arrives a few tens of microseconds into the step.
            # A short raw spin, then yields, then 20 us sleeps (a wait can last a whole step, and a core spinning
            # per rank competes with the launch threads of the other ranks of a node).
            t0, spins = time.perf_counter(), 0
            while int(tail[3]) != seq:
                spins += 1
                if spins <= 64:
                    continue
                waited = time.perf_counter() - t0
                if waited > 5.0:               # (never in a healthy run: let the runtime say what went wrong)
                    torch.cuda.synchronize(self.inputs[1].device)
                    if int(tail[3]) != seq:
                        raise hip.HipLibraryError('the deferred step never reported its edge count (sequence number missing)')
                    break
                time.sleep(0 if waited < 50e-6 else 20e-6)
        self.count, self.bad, changed = int(tail[0]), int(tail[1]), int(tail[2])
        if changed:                        # (the prepared block's change counter: a parameter differs from its snapshot)
            self.bad |= hip.STATUS_PARAMS_CHANGED
        self.tail = self.event = None
        self.state = _Deferred.WORDS
        # a step that ran on an emptied graph / a stale block / the wrong kernel forms is worth nothing until it is repeated: count
        # them, whether or not anybody touches the result (bench.py reports the count next to its timings)
        if (self.count > self.cap or (self.bad & (hip.STATUS_PARAMS_CHANGED | 3))
                or self.small_molecules != (not (self.bad & hip.STATUS_BIG_MOLECULE))):
            st = self.owner.__dict__.setdefault('_deferred_stats', {'deferred_calls': 0, 'repeats_needed': 0})
            st['repeats_needed'] += 1
        if self.bad & 1:
            self.error = ValueError('batch must be non-decreasing with values in [0, cell.shape[0]) (PyG collation order)')
        elif self.bad & 2:
            self.error = IndexError('atomic numbers z must lie in [0, 118] (rows of node_embedding / scale / shift)')
        else:
            self.owner._note_count(self.inputs[1].shape[0], self.count, self.inputs[2].shape[0], self.bad)

    def settle(self):
        if self.state == _Deferred.DONE:
            if self.error is not None:
                raise self.error
            return self
        with _CallGuard(self.owner, self.inputs[1].device):     # (a repeat touches the module's workspace: same rules as a call)
            return self._settle_locked()

    def _settle_locked(self):
        if self.state == _Deferred.DONE:                        # (another thread settled it while this one waited for the lock)
            if self.error is not None:
                raise self.error
            return self
        self.read_words()
        if self.error is not None:
            self.state, self.reported = _Deferred.DONE, True
            raise self.error
        # (a call queued with the wrong guess about molecule sizes is repeated as well: its kernels were correct, but the
        # synchronous path would have summed force_fwd's rows in another order -- results never depend on the path taken)
        if (self.count > self.cap or (self.bad & hip.STATUS_PARAMS_CHANGED)
                or self.small_molecules != (not (self.bad & hip.STATUS_BIG_MOLECULE))):
            if _versions(self.inputs) != self.versions:
                self.error = RuntimeError(
                    'this forward call has to be repeated (its edge count exceeded the capacity taken from the previous call, or a '
                    'parameter had changed), but one of its input tensors was modified in place before its outputs were read: '
                    'read the outputs before modifying the inputs, or call model.synchronize_checks() right after forward')
                self.state = _Deferred.DONE
                raise self.error
            owner = self.owner
            z, pos, cell, batch = self.inputs
            with torch.no_grad():
                model = owner._hip_model(self.energy_idx)
                if self.param_stamp is not None and owner.__dict__.get('_param_stamp') != self.param_stamp:
                    self.error = RuntimeError(
                        'this forward call has to be repeated (its edge count exceeded the capacity taken from the previous call, or '
                        'its guess about the molecule sizes was wrong), but a parameter of the module was modified in place before '
                        'its outputs were read: read the outputs before updating the parameters, or call model.synchronize_checks() '
                        'right after forward')
                    self.state = _Deferred.DONE
                    raise self.error
                self.res, self.graph = owner._forward_sync(model, self.zc, pos, cell, batch, self.want_forces, self.want_virial)
        else:
            self.graph.n_edges, self.graph.status = self.count, self.bad
        self.state = _Deferred.DONE
        return self

    def result(self, name):
        return self.settle().res[name]


class EmbeddingNet(nn.Module):
    """Node embedding table + edge-embedding hyper-parameters (newtonnet.py:116-137)."""
    def __init__(self, cutoff, n_features, n_basis):
        super().__init__()
        self.n_features = n_features
        self.node_embedding = nn.Embedding(118 + 1, n_features, padding_idx=0)
        self.edge_embedding = EdgeEmbedding(cutoff=cutoff, n_basis=n_basis)
        self.requires_dr = False


class InteractionNet(nn.Module):
    """Parameters of one message-passing layer (newtonnet.py:175-205)."""
    def __init__(self, n_features, n_basis, activation, layer_norm):
        super().__init__()
        self.n_features = n_features
        self.message_nodepart = nn.Sequential(nn.Linear(n_features, n_features), activation,
                                              nn.Linear(n_features, n_features))
        self.message_edgepart = nn.Linear(n_basis, n_features, bias=False)
        self.equiv_message1 = nn.Sequential(nn.Linear(n_features, n_features, bias=False), activation,
                                            nn.Linear(n_features, n_features, bias=False))
        self.equiv_message2 = nn.Sequential(nn.Linear(n_features, n_features, bias=False), activation,
                                            nn.Linear(n_features, n_features, bias=False))
        self.equiv_update = nn.Linear(n_features, n_features, bias=False)
        self.layer_norm = nn.LayerNorm(n_features) if layer_norm else None


class NewtonNet(nn.Module):
    """Molecular Newtonian message passing, MI355X-native.

    Parameters (identical to the reference, newtonnet.py:26-35):
        cutoff, n_features, n_basis, n_interactions, activation, layer_norm, output_properties
    """
    def __init__(self, cutoff: float = 5.0, n_features: int = 128, n_basis: int = 20, n_interactions: int = 3,
                 activation: str = 'swish', layer_norm: bool = False, output_properties: list = []) -> None:
        super().__init__()
        self.activation_name = activation
        act = get_activation_by_string(activation)
        self.embedding_layers = EmbeddingNet(cutoff=cutoff, n_features=n_features, n_basis=n_basis)
        self.interaction_layers = nn.ModuleList([
            InteractionNet(n_features=n_features, n_basis=n_basis, activation=act, layer_norm=layer_norm)
            for _ in range(n_interactions)])
        self.output_properties = output_properties
        self.output_layers = nn.ModuleList()
        self.scalers = nn.ModuleList()
        self.aggregators = nn.ModuleList()
        for key in self.output_properties:
            output_layer = get_output_by_string(key, n_features, act)
            self.output_layers.append(output_layer)
            if isinstance(output_layer, DerivativeProperty):
                self.embedding_layers.requires_dr = True
            self.scalers.append(get_scaler_by_string(key))
            self.aggregators.append(get_aggregator_by_string(key))

    # ------------------------------------------------------------------------------------------
    # per-module run-time state of the HIP path (workspaces, capacity hints, pinned slots, the record of the last queued call ...):
    # never pickled, never shared between lanes
    _RUNTIME_KEYS = ('_train_ws', '_static_train_graph', '_infer_ws', '_prep_block', '_edge_hint', '_mol_hint', '_param_stamp',
                     '_param_epoch', '_tail_ring', '_last_deferred', '_force_sync', '_model_cache', '_deferred_stats', '_call_lock',
                     '_last_stream', '_lanes', '_lane_of')

    def __getstate__(self):
        """Whole-module pickles (trainer.py:219) carry parameters and structure only: the training workspaces stay behind."""
        state = self.__dict__.copy()
        for k in self._RUNTIME_KEYS:
            state.pop(k, None)
        return state

    def inference_lanes(self, n=2):
        """`n` views of this module for `n` evaluation steps IN FLIGHT at once, each driven from its own HIP stream:

            lanes, streams = model.inference_lanes(2), [torch.cuda.Stream() for _ in range(2)]
            for k, (z, pos, cell, batch) in enumerate(batches):           # independent batches (conformer screening, test sets)
                with torch.cuda.stream(streams[k % 2]):
                    outs.append(lanes[k % 2](z, pos, cell, batch))

        (The usual multi-stream rules apply: inputs produced on another stream need `streams[k].wait_stream(producer)` first, inputs
        freed right after the call `record_stream`, and a lane's results are consumed on the stream that produced them.)
        One module serialises its calls on the GPU (they share one workspace); a step of this path is a chain of 30-45 dependent
        launches whose fill / drain phases leave the chip partly idle, and a second, independent step on another stream runs in
        those gaps: 1024 aspirin conformers 1.50 -> 1.38 ms per step, 128 conformers 353 -> 263 us (profiles/r06_two_stream.txt).
        A lane is a shallow copy: the SAME Parameter and submodule objects (load_state_dict / .to() / an optimizer step on the owner
        reach every lane), its own run-time state (workspace ~1.7 GB at config-2 size, prepared block, capacity hints, deferred
        checks, lock).  Lane 0 is the module itself.  Lanes are eval-only.  The reference has one eager path and no counterpart
        (newtonnet/models/newtonnet.py:74-104)."""
        import copy
        owner = self.__dict__.get('_lane_of')
        if owner is not None:
            return owner.inference_lanes(n)
        lanes = self.__dict__.setdefault('_lanes', [self])
        while len(lanes) < n:
            lane = copy.copy(self)             # (__getstate__ drops the run-time state; _parameters / _modules are the same objects)
            lane.__dict__['_lane_of'] = self
            lane.training = False
            lanes.append(lane)
        return lanes[:n]

    # ------------------------------------------------------------------------------------------
    def train(self, mode=True):
        """As the reference (newtonnet.py:106-113): flips create_graph on derivative heads; returns None."""
        if '_lane_of' in self.__dict__:     # an inference lane: its submodules are the OWNER's -- never touch their flags from here
            if mode:
                raise RuntimeError('an inference lane is eval-only (its submodules are the owner\'s): call train() on the owner module')
            self.training = False
            return
        super().train(mode)
        for output_layer in self.output_layers:
            if isinstance(output_layer, DerivativeProperty):
                output_layer.create_graph = mode

    # ------------------------------------------------------------------------------------------
    def _hip_model(self, energy_idx: int) -> hip.Model:
        """The nnhip_model struct of this module (device addresses of every parameter).  Built by a full walk with all checks
        (`_hip_model_build`) and kept with what is needed to see, in a few microseconds, that it still describes the module:
        the identity of every submodule on the way to a parameter (`module._modules[name] is child`), and address / dtype /
        contiguity of every parameter tensor read straight from the `_parameters` dicts.  Anything else -- a replaced
        submodule or Parameter, `.to()` / `.data =` swaps, head surgery on output_layers / scalers -- rebuilds it.  Only
        ADDRESSES are cached; parameter VALUES are compared on the device every call (nnhip_prepare_check)."""
        c = self.__dict__.get('_model_cache')
        if c is not None and c[0] == energy_idx:
            ee = self.embedding_layers.edge_embedding
            if c[1] == (self.activation_name, ee.cutoff, ee.n_basis, ee.envelope_id, len(self.interaction_layers._modules)):
                ok = True
                for d, k, child in c[2]:
                    if d.get(k) is not child:
                        ok = False
                        break
                if ok:
                    f32 = torch.float32
                    vsum = 0
                    for (d, k), ptr in zip(c[3], c[4]):
                        t = d[k]
                        if t.data_ptr() != ptr or t.dtype is not f32 or not t.is_contiguous():
                            ok = False
                            break
                        vsum += t._version
                    if ok:
                        # (epoch of the cached struct, sum of the in-place version counters: what a deferred call that has to be
                        # repeated compares, _Deferred.settle -- the parameter VALUES it ran on are gone once somebody wrote to them)
                        self.__dict__['_param_stamp'] = (c[6], vsum)
                        return c[5]
        m = self._hip_model_build(energy_idx)
        try:
            entry = self._model_cache_entry(energy_idx, m)
            epoch = self.__dict__.get('_param_epoch', 0) + 1
            self.__dict__['_param_epoch'] = epoch
            self.__dict__['_model_cache'] = entry + (epoch,)
            self.__dict__['_param_stamp'] = (epoch, sum(d[k]._version for d, k in entry[3]))
        except (KeyError, AttributeError):      # an unexpected module layout: no cache, the full walk serves every call
            self.__dict__.pop('_model_cache', None)
            self.__dict__['_param_stamp'] = None
        return m

    def _model_cache_entry(self, energy_idx, m):
        guards, seen, params, ptrs = [], set(), [], []

        def add(*path):
            mod = self
            for name in path[:-1]:
                # (an absent optional submodule -- layer_norm / scale / shift = None -- is a plain None attribute, not an
                # entry of _modules; assigning a Module later creates the entry, which the guard below then sees)
                child = mod._modules.get(name)
                if child is None and getattr(mod, name) is not None:
                    raise KeyError(name)
                if (id(mod), name) not in seen:
                    seen.add((id(mod), name))
                    guards.append((mod._modules, name, child))
                if child is None:
                    return
                mod = child
            t = mod._parameters[path[-1]]
            params.append((mod._parameters, path[-1]))
            ptrs.append(t.data_ptr())

        add('embedding_layers', 'node_embedding', 'weight')
        add('embedding_layers', 'edge_embedding', 'embedding', 'frequencies')
        ee = self.embedding_layers.edge_embedding
        guards.append((ee._modules, 'envelope', ee._modules.get('envelope')))
        for l in range(len(self.interaction_layers)):
            il = ('interaction_layers', str(l))
            for sub in (('message_nodepart', '0', 'weight'), ('message_nodepart', '0', 'bias'), ('message_nodepart', '2', 'weight'),
                        ('message_nodepart', '2', 'bias'), ('message_edgepart', 'weight'), ('equiv_message1', '0', 'weight'),
                        ('equiv_message1', '2', 'weight'), ('equiv_message2', '0', 'weight'), ('equiv_message2', '2', 'weight'),
                        ('equiv_update', 'weight'), ('layer_norm', 'weight'), ('layer_norm', 'bias')):
                add(*il, *sub)
        k = str(energy_idx)
        for sub in (('0', 'weight'), ('0', 'bias'), ('2', 'weight'), ('2', 'bias'), ('4', 'weight'), ('4', 'bias')):
            add('output_layers', k, 'layers', *sub)
        add('scalers', k, 'scale', 'weight')
        add('scalers', k, 'shift', 'weight')
        key = (self.activation_name, ee.cutoff, ee.n_basis, ee.envelope_id, len(self.interaction_layers._modules))
        return (energy_idx, key, guards, params, ptrs, m)

    def _hip_model_build(self, energy_idx: int) -> hip.Model:
        emb = self.embedding_layers
        F, nb, L = emb.n_features, emb.edge_embedding.n_basis, len(self.interaction_layers)
        if F != hip.NNHIP_F or not (1 <= nb <= hip.NNHIP_MAX_NB) or not (1 <= L <= hip.NNHIP_MAX_LAYERS):
            raise NotImplementedError(f'HIP kernels are built for n_features={hip.NNHIP_F}, n_basis<={hip.NNHIP_MAX_NB}, '
                                      f'1..{hip.NNHIP_MAX_LAYERS} interactions (got {F}, {nb}, {L})')
        if self.activation_name not in HIP_FUSED:
            raise NotImplementedError(f"activation '{self.activation_name}' is not fused into the HIP kernels")

        def p(t):
            if t.dtype != torch.float32 or not t.is_cuda:
                raise NotImplementedError(f'HIP path needs float32 parameters on the GPU (got {t.dtype} on {t.device}); '
                                          f'call model.to(torch.float32).to("cuda")')
            if not t.is_contiguous():
                raise ValueError('non-contiguous parameter')
            return t.data_ptr()

        m = hip.Model()
        m.n_features, m.n_basis, m.n_layers = F, nb, L
        m.cutoff = float(emb.edge_embedding.cutoff)
        m.node_embedding = p(emb.node_embedding.weight)
        m.frequencies = p(emb.edge_embedding.embedding.frequencies)
        for l, il in enumerate(self.interaction_layers):
            lp = m.layer[l]
            lp.node0_w, lp.node0_b = p(il.message_nodepart[0].weight), p(il.message_nodepart[0].bias)
            lp.node2_w, lp.node2_b = p(il.message_nodepart[2].weight), p(il.message_nodepart[2].bias)
            lp.edge_w = p(il.message_edgepart.weight)
            lp.eq1_0_w, lp.eq1_2_w = p(il.equiv_message1[0].weight), p(il.equiv_message1[2].weight)
            lp.eq2_0_w, lp.eq2_2_w = p(il.equiv_message2[0].weight), p(il.equiv_message2[2].weight)
            lp.update_w = p(il.equiv_update.weight)
            if il.layer_norm is not None:      # newtonnet.py:202-205,228-231 (eps = nn.LayerNorm default 1e-5)
                if abs(il.layer_norm.eps - 1e-5) > 0 or not il.layer_norm.elementwise_affine:
                    raise NotImplementedError('HIP path implements nn.LayerNorm(n_features) with its default eps / affine')
                lp.ln_w, lp.ln_b = p(il.layer_norm.weight), p(il.layer_norm.bias)
        head = self.output_layers[energy_idx].layers
        m.head0_w, m.head0_b = p(head[0].weight), p(head[0].bias)
        m.head2_w, m.head2_b = p(head[2].weight), p(head[2].bias)
        m.head4_w, m.head4_b = p(head[4].weight), p(head[4].bias)
        sc = self.scalers[energy_idx]
        m.scale = p(sc.scale.weight) if sc.scale is not None else None
        m.shift = p(sc.shift.weight) if sc.shift is not None else None
        m.activation = hip.ACTIVATION_IDS[self.activation_name]
        m.envelope = emb.edge_embedding.envelope_id
        return m

    # ------------------------------------------------------------------------------------------
    def forward(self, z, pos, cell, batch):
        """Network forward pass (newtonnet.py:74-104).

        z int64 [N]; pos float32 [N,3]; cell float32 [B,3,3] (all-zero = non-periodic); batch int64 [N], sorted.
        Returns a CustomOutputSet with z, pos, atom_node, force_node, edge_index, cell, displacement, batch and one
        attribute per output property (`energy` [B], `gradient_force` [N,3], ...).
        """
        if not pos.is_cuda:
            raise RuntimeError('newtonnet_amd.NewtonNet runs on an MI355X (ROCm) device only: move the model and the '
                               'inputs to "cuda".  There is no CPU path in this package.')
        keys = list(self.output_properties)
        if 'energy' not in keys:
            raise NotImplementedError("the HIP hot path needs the 'energy' head (output_properties)")
        for key in keys:
            if key not in ('energy', 'gradient_force', 'direct_force', 'virial', 'stress'):
                raise NotImplementedError(f"output property '{key}' is outside the MI355X hot path")
        deriv_layers = [ol for ol in self.output_layers if isinstance(ol, DerivativeProperty)]
        # differentiable path: whenever the module is in train mode with autograd on -- with or without a derivative head
        # (the reference trains ['energy'] or ['energy', 'direct_force'] models just the same, trainer.py:299-313)
        # (an inference lane shares its submodules -- and their create_graph flags -- with its owner, and is eval-only whatever the owner does)
        train_graph = (torch.is_grad_enabled() and (self.training or any(ol.create_graph for ol in deriv_layers))
                       and '_lane_of' not in self.__dict__)
        energy_idx = keys.index('energy')
        want_forces = len(deriv_layers) > 0
        want_virial = any(isinstance(ol, (VirialOutput, StressOutput)) for ol in deriv_layers)

        emb = self.embedding_layers
        # the reference marks the caller's tensor (newtonnet.py:150-152); kept for API parity
        mark = emb.requires_dr and pos.is_leaf and pos.is_floating_point()
        if mark:
            pos.requires_grad = True

        def make_displacement():
            d = torch.eye(3, dtype=pos.dtype, device=pos.device).repeat(cell.shape[0], 1, 1)
            if mark:
                d.requires_grad = True
            return d

        if train_graph:
            return self._forward_train(z, pos, cell, batch, keys, energy_idx, make_displacement())

        with torch.no_grad(), _CallGuard(self, pos.device):
            # bookkeeping of the previous deferred call (edge-count hint, stale prepared block); raises ITS deferred error when
            # nobody has looked at its outputs yet
            self._settle_last()
            model = self._hip_model(energy_idx)
            # one big system: the kernels see the atoms in Morton order of cells (hip.spatial_order); every per-atom result below
            # goes back through the inverse permutation (hip.permute_rows), the neighbor list through hip.edge_index_unpermute
            order_min = self.__dict__.get('_spatial_order_min', _SPATIAL_ORDER_MIN)
            perm = inv = None
            z_in, pos_in = z, pos
            if order_min > 0 and cell.shape[0] == 1 and pos.shape[0] >= order_min and pos.dtype == torch.float32:
                z64 = z.contiguous() if z.dtype == torch.int64 else z.long().contiguous()
                perm, inv, z_in, pos_in = hip.spatial_order(pos.detach(), z64, emb.edge_embedding.cutoff)
            zc = z_in.contiguous() if z_in.dtype == torch.int64 else z_in.long().contiguous()
            rec = _Deferred(self, (z_in, pos_in, cell, batch), zc, energy_idx, want_forces, want_virial)
            rec.param_stamp = self.__dict__.get('_param_stamp')
            if not self._forward_deferred(rec, model):
                rec.res, rec.graph = self._forward_sync(model, zc, pos_in, cell, batch, want_forces, want_virial)
                rec.state = _Deferred.DONE

        def own_order(t):       # a per-atom result of the call, in the caller's order
            return t if perm is None else hip.permute_rows(t, inv)

        def own_edges():        # the neighbor list as the reference lists it for the caller's order
            g = rec.settle().graph
            return hip.edge_index_unpermute(g.row_ptr, g.col, perm, inv, pos.shape[0], int(g.n_edges))

        # Every result of the call sits behind the record: touching one settles the deferred host-side checks first (a few
        # microseconds when the words have arrived, which they have unless the host is a whole step ahead of the GPU).
        outputs = CustomOutputSet(z=z, pos=pos, cell=cell, batch=batch)
        outputs.lazy('displacement', make_displacement)
        outputs.lazy('atom_node', lambda: own_order(rec.result('atom_node')))
        outputs.lazy('force_node', lambda: own_order(rec.result('force_node')))
        # (the directed-edge count alone -- not an attribute of the reference's bag: callers that want E need not build [2][E] int64)
        outputs.lazy('n_edges', lambda: int(rec.settle().graph.n_edges))
        if perm is None:
            outputs.lazy('edge_index', lambda: rec.settle().graph.edge_index)
        else:
            outputs.lazy('edge_index', own_edges)
        if want_forces:
            outputs.lazy('pos_grad', lambda: -own_order(rec.result('forces')))
            if want_virial:
                outputs.lazy('displacement_grad', lambda: -rec.result('virial'))
        for key in keys:
            if key == 'energy':
                outputs.lazy('energy', lambda: rec.result('energy'))
            elif key == 'gradient_force':
                outputs.lazy('gradient_force', lambda: own_order(rec.result('forces')))
            elif key == 'virial':
                outputs.lazy('virial', lambda: rec.result('virial'))
            elif key == 'stress':
                outputs.lazy('stress', lambda: -rec.result('virial') / cell.det().view(-1, 1, 1))
            elif key == 'direct_force':
                def direct(k=keys.index('direct_force')):
                    sc = self.scalers[k].scale
                    with torch.no_grad():
                        return own_order(hip.direct_force(rec.result('atom_node'), rec.result('force_node'), zc,
                                                          self.output_layers[k].layers, sc.weight if sc is not None else None,
                                                          hip.ACTIVATION_IDS[self.activation_name]))
                outputs.lazy('direct_force', direct)
        if rec.state == _Deferred.DONE:      # (synchronous call: nothing is pending, hand the tensors over as plain attributes)
            # (a re-ordered big system keeps its node states and its neighbor list lazy: their copies back into the caller's order
            # are 200 MB and a 5 M-key sort at 100k atoms, and most callers read energy and forces only)
            eager = ('energy', 'gradient_force', 'virial') if perm is not None else \
                ('atom_node', 'force_node', 'edge_index', 'energy', 'gradient_force', 'virial')
            for name in eager:
                if name in outputs.__dict__.get('_lazy', {}):
                    getattr(outputs, name)
        return outputs

    # ------------------------------------------------------------------------------------------
    def _prep_key(self, model, device):
        return (device, model.n_layers, model.n_basis, model.activation, model.envelope, hip.lib().nnhip_split_products())

    def _forward_sync(self, model, zc, pos, cell, batch, want_forces, want_virial):
        """One eval-mode call with the host in the loop: the edge count (and the status word) are read back before the
        per-edge arrays are sized -- the first call of a module, the first call with a new atom count, periodic boxes on the
        cell-list builder, and the repeat of a deferred call whose capacity did not fit or whose prepared block was stale.
        Returns (result dict, Graph)."""
        emb = self.embedding_layers
        # Parameter-only preparation (transposed weights, split-f16 weight images, radial-filter tables, layer 0's
        # message_nodepart per element) lives in one block per module and is refilled only when a parameter CHANGED: every
        # call compares the parameters bit for bit with the snapshot nnhip_prepare took (nnhip_prepare_check, one small launch
        # ahead of the edge-count read-back; compare only -- a change stays visible until the block HAS been refilled, so a
        # call that fails in between cannot lose it) and the answer comes back with the count.  Nothing is keyed on tensor
        # identity or version counters.  NNHIP_PREPARE_EVERY_CALL=1 refills the block on every call (A/B timing).
        key = self._prep_key(model, pos.device)
        cached = self.__dict__.get('_prep_block')
        fresh = cached is None or cached[0] != key
        block = None if fresh else cached[1]
        refill = fresh or _PREPARE_EVERY_CALL
        prep = []

        def before_sync(status):
            if not refill:
                hip.prepare_check(model, block, status)

        def in_the_bubble():   # the host allocates what does not depend on the edge count while it waits for it
            if refill:
                prep.append(hip.prepare(model, pos.device, block))
            prep.append(hip.alloc_outputs(pos.shape[0], cell.shape[0], pos.device, want_forces, want_virial))
        # The neighbor-list fill is queued BEFORE the host has the edge count, into arrays sized from the previous call with
        # the same atom count: its kernels read the count on the device and write nothing when it does not fit (the ordinary
        # path then runs after the wait).  NNHIP_GRAPH_EARLY=0: always the ordinary path (A/B timing).
        hint = self.__dict__.get('_edge_hint', (None, 0))
        cap = hint[1] if (hint[0] == pos.shape[0] and _GRAPH_EARLY) else 0
        g = hip.build_graph(pos.detach(), cell.detach(), batch, emb.edge_embedding.cutoff,
                            emb.edge_embedding.embedding.frequencies,
                            while_waiting=in_the_bubble, before_sync=before_sync,
                            z=zc, envelope=emb.edge_embedding.envelope_id, edge_capacity=cap)
        self._note_count(pos.shape[0], g.n_edges, cell.shape[0], g.status)
        if refill:
            block = prep[0]
            self.__dict__['_prep_block'] = (key, block)
        elif g.status & hip.STATUS_PARAMS_CHANGED:
            hip.prepare(model, pos.device, block)
        # the workspace of the previous call is reused when it is large enough (the arrays in it are private to one call;
        # everything the caller sees lives in the output tensors)
        res = hip.energy_forces(model, zc, pos.detach(), cell.detach(), g, want_forces=want_forces,
                                want_virial=want_virial, prepared=block, out=prep[-1], workspace=self.__dict__.get('_infer_ws'))
        self.__dict__['_infer_ws'] = res['workspace']
        return res, g

    def _note_count(self, n_atoms, n_edges, n_mol=0, status=0):
        """Capacity of the per-edge arrays for the next call with this atom count: the last count + 1/16 (+ 256), kept while
        the counts stay inside it with some room to spare (a stable capacity = stable allocation sizes).  Also what the count pass
        said about molecule sizes: while batches of this (atoms, molecules) shape hold no molecule above NNHIP_MOL_STAGE_MAX
        atoms, the next deferred call may launch the molecule-resident edge kernels (they stay correct on a larger molecule,
        only slow; the status word of that call switches them off again)."""
        self.__dict__['_mol_hint'] = (n_atoms, n_mol, not (status & hip.STATUS_BIG_MOLECULE))
        hint = self.__dict__.get('_edge_hint', (None, 0))
        if n_edges <= 0:
            cap = 0
        elif hint[0] == n_atoms and n_edges + (n_edges >> 5) <= hint[1] <= n_edges + (n_edges >> 3) + 512:
            cap = hint[1]
        else:
            cap = (n_edges + (n_edges >> 4) + 256 + 1) & ~1
        self.__dict__['_edge_hint'] = (n_atoms, cap)

    def _forward_deferred(self, rec, model):
        """The steady-state call: NOTHING waits for the device.  The neighbor list, the edge embedding and the whole
        energy / force pipeline are queued into arrays sized from the previous call's edge count (the kernels read the true
        count on the device); the (count, status) words travel to pinned host memory on the side and are looked at when a result
        of the call is first touched or when the next call starts (`_Deferred.settle`).  Returns False when the call has to
        take the synchronous path (no capacity yet for this atom count, no prepared block yet, a box the cell-list builder
        serves, NNHIP_DEFERRED=0 in the environment, or `model.deferred_checks = False` on this module: every call then
        raises its data-dependent errors on the spot, as the reference does)."""
        z, pos, cell, batch = rec.inputs
        N, B = pos.shape[0], cell.shape[0]
        hint = self.__dict__.get('_edge_hint', (None, 0))
        cached = self.__dict__.get('_prep_block')
        if (not _DEFERRED or not self.__dict__.get('deferred_checks', True) or _PREPARE_EVERY_CALL or hint[0] != N or hint[1] < 2
                or cached is None
                or cached[0] != self._prep_key(model, pos.device) or (B == 1 and N >= hip.CELL_LIST_MIN_ATOMS)
                or self.__dict__.pop('_force_sync', False)):
            return False
        block, cap = cached[1], hint[1]
        # pinned (count, status, change counter, seq) slots: a ring of four per module (a slot is read by the next call at the
        # latest, _settle_last).  No event: the last neighbor-list kernel stores the call's sequence number behind the three
        # words and the host polls for it (_Deferred.read_words) -- an event record is a marker packet, ~6 us of bubble per step.
        ring = self.__dict__.get('_tail_ring')
        if ring is None or ring[3] != pos.device:
            tails = torch.zeros(4, 4, dtype=torch.int32).pin_memory()
            ring = self.__dict__['_tail_ring'] = [tails, tails.numpy(), 0, pos.device, tails.data_ptr()]
        ring[2] = ring[2] % 0x3fffffff + 1           # (never 0: the slots start zeroed)
        seq = ring[2]
        k = seq & 3
        pd, cd = pos.detach(), cell.detach()
        if pd.dtype != torch.float32 or cd.dtype != torch.float32:
            raise NotImplementedError(f'the HIP path computes in float32 (got pos {pd.dtype}, cell {cd.dtype})')
        bt = batch if (batch.dtype == torch.int64 and batch.is_contiguous()) else batch.long().contiguous()
        small = rec.small_molecules = self.__dict__.get('_mol_hint') == (N, B, True)
        st = hip.forward_dev(model, rec.zc, pd.contiguous(), cd.contiguous(), bt, cap, block, ring[4] + 16 * k, seq,
                             rec.want_forces, rec.want_virial, self.__dict__.get('_infer_ws'),
                             small_molecules=small)
        self.__dict__['_infer_ws'] = st.workspace
        rec.res, rec.graph, rec.tail, rec.event, rec.cap = st, st, ring[1][k], seq, cap
        rec.versions = _versions(rec.inputs)
        rec.state = _Deferred.QUEUED
        self.__dict__['_last_deferred'] = rec
        self.__dict__.setdefault('_deferred_stats', {'deferred_calls': 0, 'repeats_needed': 0})['deferred_calls'] += 1
        return True

    def _settle_last(self):
        rec = self.__dict__.pop('_last_deferred', None)
        if rec is not None and rec.state == _Deferred.QUEUED:
            rec.read_words()
            if rec.bad & hip.STATUS_PARAMS_CHANGED:      # the block is stale: this call takes the synchronous path, which refills it
                self.__dict__['_force_sync'] = True
            if rec.error is not None and not rec.reported:
                rec.reported = True
                raise type(rec.error)(f'(raised by the PREVIOUS forward call, whose checks were deferred) {rec.error}')

    def deferred_stats(self, reset=False):
        """{'deferred_calls': eval calls queued without waiting for the device, 'repeats_needed': how many of them ran on an emptied
        graph, a stale prepared block, a wrong molecule-size guess or invalid inputs -- i.e. produced nothing usable until repeated}.
        The last queued call is settled first."""
        with self._module_lock():      # (another thread may be settling the same record: count a repeat once)
            rec = self.__dict__.get('_last_deferred')
            if rec is not None:
                rec.read_words()
            st = dict(self.__dict__.setdefault('_deferred_stats', {'deferred_calls': 0, 'repeats_needed': 0}))
            if reset:
                self.__dict__['_deferred_stats'] = {'deferred_calls': 0, 'repeats_needed': 0}
        return st

    def synchronize_checks(self):
        """Settle the deferred host-side checks of the last eval-mode call now (raises what it would have raised)."""
        with self._module_lock():
            rec = self.__dict__.get('_last_deferred')
            if rec is not None:
                rec.settle()

    def _module_lock(self):
        """The per-module re-entrant lock of _CallGuard (created on first use)."""
        d = self.__dict__
        lock = d.get('_call_lock')
        if lock is None:
            with _LOCK_CREATION:
                lock = d.setdefault('_call_lock', threading.RLock())
        return lock

    # ------------------------------------------------------------------------------------------
    def _forward_train(self, z, pos, cell, batch, keys, energy_idx, displacement):
        """Train mode (create_graph=True): outputs stay attached to autograd so a force loss can be back-propagated
        (trainer.py:301-313): newtonnet_amd/train_fused.py, no torch autograd graph inside the step."""
        from newtonnet_amd import train_fused
        for key in keys:
            if key not in ('energy', 'gradient_force', 'direct_force'):
                raise NotImplementedError(f"train-mode forward supports energy / gradient_force / direct_force (got '{key}')")
        self._hip_model(energy_idx)          # same support checks as the inference path (fp32, F=128, SiLU, ...)
        if 'gradient_force' in keys and not pos.requires_grad:
            raise RuntimeError('train-mode forward needs pos to be a leaf tensor that can require grad')
        if not train_fused.supported(self, keys):
            raise NotImplementedError('train-mode forward: LayerNorm on some interaction layers only (or a repeated output key) has no '
                                      'hand-written training path')
        static = getattr(self, '_static_train_graph', None)
        # ONE autograd node on the hand-written kernels (tangent-over-reverse, csrc/train.hip; direct_force head csrc/heads.hip;
        # LayerNorm value / tangent kernels between unfused node stages)
        energy, forces, direct, g, ws = train_fused.forward_train(self, z, pos, cell, batch, graph=static)
        outputs = CustomOutputSet(z=z, pos=pos, edge_index=g.edge_index, cell=cell, displacement=displacement, batch=batch)
        outputs.lazy('atom_node', lambda: ws.a_out[-1].clone())
        outputs.lazy('force_node', lambda: ws.f_out[-1].clone())
        outputs.energy = energy
        if 'gradient_force' in keys:
            outputs.gradient_force = forces
            outputs.lazy('pos_grad', lambda: -forces)
        if 'direct_force' in keys:
            outputs.direct_force = direct
        return outputs
