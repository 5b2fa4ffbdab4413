"""Train-mode forward of the hot path on the hand-written kernels: (theta, pos) -> (energy, forces) as ONE autograd node.

The reference trains by back-propagating through the autograd force (`create_graph=True`, newtonnet/models/output.py:66-73;
newtonnet/train/trainer.py:299-313; loss newtonnet/train/loss.py:48,72,96).  Here nothing records a graph: `FusedEnergyForces`
runs the value sweeps (forward + analytic force adjoint) stage by stage through the C ABI with every intermediate kept, and its
backward -- given dL/dE and dL/dF of ANY loss -- returns the parameter gradients as "tangent over reverse":
    dL/dtheta = sum_b c_b dE_b/dtheta - D_d[grad_theta E_tot],      c = dL/dE,  d = dL/dF,
the reverse sweep differentiated once more in forward mode along v = -d (csrc/train.hip), followed by one batched launch of
split-K MFMA weight-gradient products.  Every kernel is an entry point of include/newtonnet_hip.h ("Per-stage entry points");
this file only owns buffers and order.  tests/tangent_ref.py states the same four sweeps in fp64 torch.

Supported: every trainable head set of the reference -- ['energy'], ['energy', 'gradient_force'], ['energy', 'direct_force'],
all three (trainer.py:299-313, loss.py:30-47) -- with or without layer_norm, every fused activation.  An energy-only loss is the
d = 0 case of the formula above; the direct_force head is an ordinary function of the final node states and back-propagates
once (csrc/heads.hip), its adjoint seeds entering the epsilon-part of the reverse sweep.  layer_norm=True runs the node-level
stages unfused with LayerNorm value / tangent kernels between them (csrc/train.hip).  (Round 1's torch-graph path,
train_ops.py, was removed in round 6.)  dL/dpos is not produced (None): the reference would return it, no trainer uses it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from newtonnet_amd import hip

F = hip.NNHIP_F
_vp = C.c_void_p


def _p(t: Optional[torch.Tensor], offset_floats: int = 0):
    return None if t is None else _vp(t.data_ptr() + 4 * offset_floats)


def _chk(rc, what):
    if rc != 0:
        raise hip.HipLibraryError(f'{what} failed (code {rc}): {hip.lib().nnhip_last_error().decode()}')


def trainable_parameters(model) -> List[torch.nn.Parameter]:
    """The parameters the fused node differentiates, in a fixed order (everything but the frozen Bessel frequencies)."""
    return [p for n, p in model.named_parameters() if 'frequencies' not in n]


def supported(model, keys) -> bool:
    keys = list(keys)
    ln = [il.layer_norm is not None for il in model.interaction_layers]
    return ('energy' in keys and set(keys) <= {'energy', 'gradient_force', 'direct_force'} and len(set(keys)) == len(keys)
            and (all(ln) or not any(ln)) and model.embedding_layers.n_features == F)


class TrainWorkspace:
    """Every buffer of one training step for N atoms, B molecules and UP TO `E` edges: value and tangent intermediates of all
    layers (the weight-gradient launch at the end reads them all), gradient outputs, and the device-resident problem tables.
    The edge count of real batches changes every step: the workspace is sized for a capacity and the kernels get the step's
    own counts (the pair-level problems of the weight-gradient table carry M = -1 = "this launch's pair count")."""

    def __init__(self, model, N: int, E: int, B: int, device):
        L = len(model.interaction_layers)
        P = E // 2
        self.N, self.E, self.B, self.L, self.P = N, E, B, L, P      # E, P: capacities
        self.busy = False

        def buf(*shape):
            return torch.empty(*shape, dtype=torch.float32, device=device)

        def per_layer(*shape, first=0):
            return [buf(*shape) if l >= first else None for l in range(L)]
        Pn = max(P, 1)
        # ---- values, forward
        self.a0 = buf(N, F)
        self.hn, self.m = per_layer(N, F), per_layer(N, F)
        self.msg = per_layer(Pn, F)
        self.h1, self.phi1 = per_layer(Pn, F), per_layer(Pn, F)
        self.h2, self.phi2 = per_layer(Pn, F, first=1), per_layer(Pn, F, first=1)
        self.a_mid, self.a_out = per_layer(N, F), per_layer(N, F)
        self.f_out, self.q = per_layer(N, 3, F), per_layer(N, 3, F)
        self.e1, self.e2, self.g_e2 = buf(N, F), buf(N, F), buf(N, F)
        self.atom_energy, self.energy, self.forces = buf(N), buf(B), buf(N, 3)
        # ---- values, reverse
        self.t_e1 = buf(N, F)
        self.GA = per_layer(N, F)
        self.gf = per_layer(N, 3, F)
        self.Gf = [buf(N, 3, F), buf(N, 3, F)]
        self.g_h12 = per_layer(Pn, 2 * F)
        self.t1, self.t2 = per_layer(Pn, F), per_layer(Pn, F, first=1)
        self.g_msg = per_layer(Pn, F)
        self.g_m, self.t_n = per_layer(N, F, first=1), per_layer(N, F, first=1)
        # ([L][E] / [L][E][4] packed for the step's own E inside flat capacity-sized buffers)
        self.g_x, self.g_u, self.g_d = buf(L * max(E, 1)), buf(L * max(E, 1) * 4), buf(max(E, 1), 4)
        # ---- tangents, forward
        self.v, self.tgeo = buf(N, 3), buf(max(E, 1), 4)
        self.da_mid = buf(N, F)
        self.da_out = per_layer(N, F)
        self.dhn, self.dm = per_layer(N, F, first=1), per_layer(N, F, first=1)
        self.dmsg = per_layer(Pn, F)
        self.dh1, self.dphi1 = per_layer(Pn, F), per_layer(Pn, F)
        self.dh2, self.dphi2 = per_layer(Pn, F, first=1), per_layer(Pn, F, first=1)
        self.df_out, self.dq = per_layer(N, 3, F), per_layer(N, 3, F)
        self.de1, self.de2 = buf(N, F), buf(N, F)
        # ---- tangents, reverse
        self.dg_e2, self.w4row, self.scal, self.dg_e1 = buf(N, F), buf(N, F), buf(N, 4), buf(N, F)
        self.dGA, self.dgf = buf(N, F), buf(N, 3, F)
        self.dGf = [buf(N, 3, F), buf(N, 3, F)]
        self.gq, self.dgq = per_layer(N, 3, F), per_layer(N, 3, F)
        self.dg_h12 = per_layer(Pn, 2 * F)
        self.dg_h1, self.dg_h2 = per_layer(Pn, F), per_layer(Pn, F, first=1)
        self.dg_msg = buf(Pn, F)
        self.g_eps, self.dg_eps = per_layer(Pn, F), per_layer(Pn, F)
        self.dg_m, self.dg_hn = per_layer(N, F), per_layer(N, F)
        self.rb = buf(Pn, 64)
        self.zeros_nf = torch.zeros(N, F, dtype=torch.float32, device=device)
        # ---- parameter-only data rebuilt every step: transposed weights, radial-filter tables
        self.wT = [[buf(F, F) for _ in range(7)] for _ in range(L)]
        self.headT = [buf(F, F), buf(F, F)]
        n_img = hip.lib().nnhip_weight_image_bytes() // 4              # split-f16 images of every weight / transpose (csrc/node128s.hip)
        self.wimg = [[buf(n_img) for _ in range(14)] for _ in range(L)]
        self.himg = [buf(n_img) for _ in range(4)]
        n_tab = hip.lib().nnhip_filter_table_bytes() // 4
        self.ftab = [buf(n_tab) for _ in range(L)]
        # ---- gradient outputs (zero-initialised once: the parameters the path never touches keep an exact zero gradient)
        # ONE flat buffer, the per-parameter gradients are views into it: the backward hands autograd views of a single copy,
        # and a data-parallel all-reduce moves the flat tensor as it is
        self.params = trainable_parameters(model)
        self.flat_grad = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=device)
        self.grads, off = [], 0
        for p in self.params:
            self.grads.append(self.flat_grad[off:off + p.numel()].view(p.shape))
            off += p.numel()
        self.sp_scratch = buf(hip.lib().nnhip_species_scratch_bytes(F) // 4)
        # ---- layer_norm=True (newtonnet.py:202-205,228-231): what the LayerNorm stages of the four sweeps keep (csrc/train.hip)
        self.has_ln = model.interaction_layers[0].layer_norm is not None
        if self.has_ln:
            self.ln_xhat, self.ln_dxhat, self.ln_gy = per_layer(N, F), per_layer(N, F), per_layer(N, F)
            self.ln_row_w, self.ln_row_b = per_layer(N, F), per_layer(N, F)
            self.ln_rstd, self.ln_drstd = per_layer(max(N, 1)), per_layer(max(N, 1))
        # ---- direct_force head (output.py:115-132), when the model has one: forward intermediates kept for its adjoint, the
        # adjoint's rows (operands of three more weight-gradient problems / column sums) and the seeds it hands to sweep 4
        self.dfh_idx = list(model.output_properties).index('direct_force') if 'direct_force' in model.output_properties else None
        if self.dfh_idx is not None:
            self.dfh_keep, self.dfh_out = buf(3, N, F), buf(N, 3)
            self.dfh_work = torch.zeros(hip.lib().nnhip_direct_force_bwd_work_floats(N), dtype=torch.float32, device=device)
            self.dfh_seed_a, self.dfh_seed_f, self.dfh_sc4 = buf(N, F), buf(N, 3, F), buf(N, 4)
            self.dfh_sp_scratch = buf(hip.lib().nnhip_species_scratch_bytes(4) // 4)
        self._tables(model, device)

    # device-resident tables of the batched launches (pointer lists of the transposes / filter tables stay on the host)
    def _tables(self, model, device):
        L, N, P = self.L, self.N, self.P
        act = hip.ACTIVATION_IDS[model.activation_name]
        gmap = {id(p): g for p, g in zip(self.params, self.grads)}
        G = lambda prm: gmap[id(prm)]  # noqa: E731
        probs, sums = [], []
        self.wgrad_cost_rows = []

        def add(out, M, A1, B1=None, A2=None, B2=None, typ=hip.WG_PLAIN, hA=None, hB=None, dhB=None, lda1=0, lda2=0, ldb1=0,
                ldb2=0, cols32=False, ldo=0, ncols=0, a1_off=0, a2_off=0, b2_off=0):
            q = hip.WgradProblem()
            q.A1, q.B1 = _p(A1, a1_off).value, (_p(B1).value if B1 is not None else None)
            q.A2 = _p(A2, a2_off).value if A2 is not None else None
            q.B2 = _p(B2, b2_off).value if B2 is not None else None
            q.hA = _p(hA).value if hA is not None else None
            q.hB = _p(hB).value if hB is not None else None
            q.dhB = _p(dhB).value if dhB is not None else None
            q.out = _p(out).value
            q.M, q.lda1, q.lda2, q.ldb1, q.ldb2, q.ldh = M, lda1, lda2, ldb1, ldb2, 0
            q.type, q.b_cols32, q.activation, q.ldo, q.ncols = typ, 1 if cols32 else 0, act, ldo, ncols
            probs.append(q)
            # cost model of the batched launch (bench.py: roofline of wgrad_kernel): products of [M x 128]^T [M x bw], and the
            # operand arrays (each read once per launch) in floats per row
            bw = 32 if cols32 else F
            n_prod = 1 + (A2 is not None)
            b_arrays = (1 if (B1 is not None or hB is not None) else 0) + (1 if (B2 is not None or dhB is not None) else 0)
            row_floats = F * n_prod + bw * b_arrays + F * ((hA is not None) + (hB is not None and B1 is not None))
            if typ == hip.WG_ACT:
                row_floats = F * n_prod + F * (1 + (dhB is not None))       # A1 (, A2), hB (, dhB)
            self.wgrad_cost_rows.append((M, n_prod, bw, row_floats))

        def colsum(out, src, rows):
            c = hip.ColsumProblem()
            c.src, c.out, c.rows = _p(src).value, _p(out).value, rows
            sums.append(c)
        nb = model.embedding_layers.edge_embedding.n_basis
        for l, il in enumerate(model.interaction_layers):
            a_in = self.a0 if l == 0 else self.a_out[l - 1]
            if P > 0:
                for k, (seq, h, dh, t, dgh) in enumerate(((il.equiv_message1, self.h1, self.dh1, self.t1, self.dg_h1),
                                                          (il.equiv_message2, self.h2, self.dh2, self.t2, self.dg_h2))):
                    if k == 1 and l == 0:
                        continue          # layer 0: phi2 multiplies force_node == 0 (newtonnet.py:143): exact zero gradient
                    add(G(seq[2].weight), -1, self.dg_h12[l], A2=self.g_h12[l], typ=hip.WG_ACT, hB=h[l], dhB=dh[l], lda1=2 * F,
                        lda2=2 * F, a1_off=k * F, a2_off=k * F)
                    add(G(seq[0].weight), -1, dgh[l], B1=self.msg[l], A2=t[l], B2=self.dmsg[l], typ=hip.WG_TDACT, hA=h[l])
                add(G(il.message_edgepart.weight), -1, self.dg_eps[l], B1=self.rb, A2=self.g_eps[l], B2=self.rb, ldb1=64, ldb2=64,
                    b2_off=32, cols32=True, ldo=nb, ncols=nb)
            n2, n0 = il.message_nodepart[2], il.message_nodepart[0]
            if l > 0:
                add(G(n2.weight), N, self.dg_m[l], A2=self.g_m[l], typ=hip.WG_ACT, hB=self.hn[l], dhB=self.dhn[l])
                add(G(n0.weight), N, self.dg_hn[l], B1=a_in, A2=self.t_n[l], B2=self.da_out[l - 1], typ=hip.WG_TDACT, hA=self.hn[l])
            else:   # the first layer's input is the embedding: its tangent is zero, the second products vanish
                add(G(n2.weight), N, self.dg_m[l], typ=hip.WG_ACT, hB=self.hn[l])
                add(G(n0.weight), N, self.dg_hn[l], B1=a_in)
            colsum(G(n2.bias), self.dg_m[l], N)
            colsum(G(n0.bias), self.dg_hn[l], N)
            add(G(il.equiv_update.weight), 3 * N, self.dgq[l], B1=self.f_out[l], A2=self.gq[l], B2=self.df_out[l])
            if self.has_ln:
                colsum(G(il.layer_norm.weight), self.ln_row_w[l], N)
                colsum(G(il.layer_norm.bias), self.ln_row_b[l], N)
        head = model.output_layers[list(model.output_properties).index('energy')].layers
        add(G(head[2].weight), N, self.dg_e2, A2=self.g_e2, typ=hip.WG_ACT, hB=self.e1, dhB=self.de1)
        add(G(head[0].weight), N, self.dg_e1, B1=self.a_out[L - 1], A2=self.t_e1, B2=self.da_out[L - 1], typ=hip.WG_TDACT,
            hA=self.e1)
        colsum(G(head[2].bias), self.dg_e2, N)
        colsum(G(head[0].bias), self.dg_e1, N)
        colsum(G(head[4].weight), self.w4row, N)
        self.g_head4_b = G(head[4].bias)
        if self.dfh_idx is not None and N > 0:      # the direct_force head's own parameters (rows written by nnhip_direct_force_bwd)
            dh = model.output_layers[self.dfh_idx].layers
            nf = N * F
            g_d3, g_pre2, g_pre1 = self.dfh_work[:nf], self.dfh_work[nf:2 * nf], self.dfh_work[3 * nf:4 * nf]
            add(G(dh[4].weight), N, g_d3, typ=hip.WG_ACT, hB=self.dfh_keep[1])
            add(G(dh[2].weight), N, g_pre2, typ=hip.WG_ACT, hB=self.dfh_keep[0])
            add(G(dh[0].weight), N, g_pre1, B1=self.a_out[L - 1])
            colsum(G(dh[4].bias), g_d3, N)
            colsum(G(dh[2].bias), g_pre2, N)
            colsum(G(dh[0].bias), g_pre1, N)
            sc = model.scalers[self.dfh_idx]
            self.dfh_g_scale = G(sc.scale.weight) if sc.scale is not None else None
        self.n_probs, self.n_sums = len(probs), len(sums)
        # split-K chunks of the weight-gradient launch: ~256 rows each (every chunk costs a 64 KiB slab written and read back),
        # but at least 32 of them while 32 rows remain per chunk (24 problems x 32 chunks fill the chip), at most 256
        # (tools/sweep_train_chunks.sh: mixed-32 step 1.15 -> 1.05 ms, aspirin-128 2.05 -> 1.93 ms against one chunk per 32 rows)
        M = max(P, N)
        self.chunks = max(1, min(256, max((M + 255) // 256, min(32, (M + 31) // 32))))
        # (csrc/train.hip:wg_partition can give the smaller problems fewer, equally long chunks -- pad_ = rows per chunk of the
        # largest problem.  Measured at 1024 aspirin conformers: the reduction 128 -> 116 us, the main launch 1190 -> 1280 us (the
        # node-level problems' long chunks become its tail), step unchanged: off.  NNHIP_WGRAD_RPC=1 turns it on for A/B.)
        if os.environ.get('NNHIP_WGRAD_RPC') == '1':
            rpc = -(-max(P, 3 * N, 1) // self.chunks)
            for q in probs:
                q.pad_ = rpc
        arr = (hip.WgradProblem * len(probs))(*probs)
        self.prob_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        arr = (hip.ColsumProblem * len(sums))(*sums)
        self.sum_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.cs_scratch = torch.empty(max(hip.lib().nnhip_colsum_scratch_bytes(self.n_sums) // 4, 1), dtype=torch.float32,
                                      device=device)
        self.slabs = torch.empty(hip.lib().nnhip_wgrad_slab_bytes(self.n_probs, self.chunks) // 4, dtype=torch.float32,
                                 device=device)
        self._c_view(model, G)

    def wgrad_cost(self, n_pairs: int):
        """(FLOPs, operand bytes) of one batched weight-gradient launch when the step has `n_pairs` pair rows."""
        fl = by = 0
        for M, n_prod, bw, row_floats in self.wgrad_cost_rows:
            m = n_pairs if M < 0 else M
            fl += 2 * m * F * bw * n_prod
            by += 4 * m * row_floats
        return fl, by

    def _c_view(self, model, G):
        """nnhip_train_ws (include/newtonnet_hip.h): every buffer above by device pointer; the batch/graph fields are set per
        step by Runner._bind."""
        if C.sizeof(hip.TrainWs) != hip.lib().nnhip_train_ws_bytes():
            raise RuntimeError('nnhip_train_ws: the ctypes mirror and the library disagree (stale libnewtonnet_hip.so?)')
        c = self.c = hip.TrainWs()
        ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        emb = model.embedding_layers.edge_embedding
        c.n_atoms, c.n_mol, c.n_layers, c.n_basis, c.envelope = self.N, self.B, self.L, emb.n_basis, emb.envelope_id
        for name, _ in hip.TrainWs._fields_:
            v = getattr(self, name, None)
            if name in ('wT', 'wimg'):
                for l in range(self.L):
                    for k in range(len(v[l])):
                        getattr(c, name)[l][k] = ptr(v[l][k])
            elif isinstance(v, list):
                arr = getattr(c, name)
                for l, t in enumerate(v):
                    arr[l] = ptr(t)
            elif isinstance(v, torch.Tensor):
                setattr(c, name, ptr(v))
        c.probs, c.sums = ptr(self.prob_dev), ptr(self.sum_dev)
        c.n_probs, c.chunks, c.n_sums = self.n_probs, self.chunks, self.n_sums
        sc = model.scalers[list(model.output_properties).index('energy')]
        c.g_embedding = ptr(G(model.embedding_layers.node_embedding.weight))
        c.g_scale = ptr(G(sc.scale.weight)) if sc.scale is not None else None
        c.g_shift = ptr(G(sc.shift.weight)) if sc.shift is not None else None
        c.g_head4_b = ptr(self.g_head4_b)
        self.model_c, self.model_key = None, None


class Runner:
    """One batch on one workspace: values() = sweeps 1-2, grads() = sweeps 3-4 + weight-gradient products."""

    def __init__(self, model, z, pos, cell, batch, g: hip.Graph, ws: TrainWorkspace):
        self.model, self.z, self.pos, self.cell, self.batch, self.g, self.ws = model, z, pos, cell, batch, g, ws
        if g.n_atoms != ws.N or g.n_mol != ws.B or g.n_edges > ws.E:
            raise ValueError(f'workspace for N={ws.N}, B={ws.B}, E<={ws.E} cannot hold a batch with N={g.n_atoms}, '
                             f'B={g.n_mol}, E={g.n_edges}')
        self.act = hip.ACTIVATION_IDS[model.activation_name]
        self.energy_idx = list(model.output_properties).index('energy')
        # under torch.autocast(bfloat16) -- the way BASELINE configs[2] asks for bf16 -- the weight-gradient products take bf16
        # operands (fp32 prologues, fp32 accumulation); every other kernel stays exact fp32
        self.bf16 = bool(torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16) \
            or os.environ.get('NNHIP_TRAIN_BF16') == '1'
        if g.rbf is None or g.drbf is None:
            raise ValueError('the training path needs a graph built with want_rbf=True')

    @property
    def st(self):
        """the CURRENT torch stream at the time of the call (a HIP-graph capture runs on its own stream)"""
        return hip._stream(self.pos.device)

    def _bind(self):
        """The C view of this step: the workspace's table of device pointers + this batch's graph, and the model's parameters."""
        ws, g, c = self.ws, self.g, self.ws.c
        key = tuple(p.data_ptr() for p in ws.params)
        if ws.model_key != key:           # (parameters re-allocated, e.g. by model.to(): rebuild the pointer table)
            ws.model_c, ws.model_key = self.model._hip_model(self.energy_idx), key
        c.n_edges = g.n_edges
        # weight-gradient products: bf16 operands under autocast(bfloat16); otherwise fp32-grade products from three bf16 pieces
        # per operand (csrc/train.hip:wgrad_split_kernel); NNHIP_WGRAD_FORM=fp32 keeps v_mfma_f32_32x32x2_f32 (A/B, tests)
        c.bf16_wgrad = 1 if self.bf16 else (0 if os.environ.get('NNHIP_WGRAD_FORM', 'split') == 'fp32' else 2)
        for name, t in (('z', self.z), ('pos', self.pos), ('cell', self.cell), ('batch', self.batch), ('mol_ptr', g.mol_ptr),
                        ('row_ptr', g.row_ptr), ('col', g.col), ('rev', g.rev), ('pid', g.pid), ('edge_index', g.edge_index),
                        ('geo', g.geo), ('disp', g.disp), ('rbf', g.rbf), ('drbf', g.drbf), ('xg', g.xg)):
            setattr(c, name, t.data_ptr())
        # the forms of the inference step for the value sweeps (round 6): per-row pair counts, and "every molecule fits the
        # molecule-resident kernels" (bit 8 of the list's status word; a static candidate list carries neither: row forms)
        pp = getattr(g, 'pair_ptr', None)
        status = getattr(g, 'status', None)
        c.pair_ptr = pp.data_ptr() if isinstance(pp, torch.Tensor) and pp.numel() == g.n_atoms + 1 else None
        c.flags = 1 if (c.pair_ptr and isinstance(status, int) and not (status & hip.STATUS_BIG_MOLECULE)) else 0
        return C.byref(ws.model_c), C.byref(c)

    # -- sweeps 1 and 2: values (csrc/train_step.hip strings the stages together) ------------------------------------------
    def _df_head(self):
        ws, model = self.ws, self.model
        head = model.output_layers[ws.dfh_idx].layers
        sc = model.scalers[ws.dfh_idx].scale
        return head, (sc.weight if sc is not None else None)

    def values(self):
        """energy [B], gradient force [N,3] (and ws.dfh_out [N,3] when the model has a direct_force head)"""
        m, c = self._bind()
        ws = self.ws
        _chk(hip.lib().nnhip_train_values(m, c, self.st), 'nnhip_train_values')
        if ws.dfh_idx is not None and ws.N > 0:
            head, scale = self._df_head()
            _chk(hip.lib().nnhip_direct_force(_p(ws.a_out[-1]), _p(ws.f_out[-1]), _p(self.z), _p(head[0].weight), _p(head[0].bias),
                                              _p(head[2].weight), _p(head[2].bias), _p(head[4].weight), _p(head[4].bias), _p(scale),
                                              self.act, ws.N, _p(ws.dfh_keep), _p(ws.dfh_out), self.st), 'nnhip_direct_force')
        return ws.energy, ws.forces

    # -- sweeps 3 and 4: tangents, then the weight gradients ---------------------------------------------------------------
    def grads(self, g_energy: torch.Tensor, g_forces: Optional[torch.Tensor], g_direct: Optional[torch.Tensor] = None):
        """dL/dparameters into ws.flat_grad given dL/dE [B], dL/d gradient_force [N,3] (None = 0: the loss has no gradient-force
        term) and, for models with a direct_force head, dL/d direct_force [N,3] (None = 0)."""
        ws = self.ws
        if g_forces is None:
            if getattr(ws, '_zero_gf', None) is None:
                ws._zero_gf = torch.zeros(ws.N, 3, dtype=torch.float32, device=ws.energy.device)
            g_forces = ws._zero_gf
        g_forces = g_forces.reshape(ws.N, 3).to(torch.float32).contiguous()
        g_energy = g_energy.to(torch.float32).contiguous()
        m, c = self._bind()
        seed_a = seed_f = None
        if ws.dfh_idx is not None and ws.N > 0:
            if g_direct is None:      # the head exists but is not in the loss: its rows of the gradient launch are zeros
                ws.dfh_work[:4 * ws.N * F].zero_()
                if ws.dfh_g_scale is not None:
                    ws.dfh_g_scale.zero_()
            else:
                head, scale = self._df_head()
                g_direct = g_direct.reshape(ws.N, 3).to(torch.float32).contiguous()
                _chk(hip.lib().nnhip_direct_force_bwd(_p(g_direct), _p(ws.f_out[-1]), _p(self.z), _p(head[0].weight),
                                                      _p(head[2].weight), _p(head[4].weight), _p(scale), self.act, ws.N,
                                                      _p(ws.dfh_keep), _p(ws.dfh_work), _p(ws.dfh_seed_a), _p(ws.dfh_seed_f),
                                                      _p(ws.dfh_sc4), _p(ws.dfh_sp_scratch), _p(ws.dfh_g_scale), self.st),
                     'nnhip_direct_force_bwd')
                seed_a, seed_f = ws.dfh_seed_a, ws.dfh_seed_f
        _chk(hip.lib().nnhip_train_grads_seeded(m, c, _p(g_energy), _p(g_forces), _p(seed_a), _p(seed_f), self.st),
             'nnhip_train_grads_seeded')
        return ws.grads


class FusedEnergyForces(torch.autograd.Function):
    """(pos, *parameters) -> (energy [B], gradient force [N,3], direct force [N,3] or an empty tensor); backward returns
    dL/dparameters (dL/dpos is not produced)."""

    @staticmethod
    def forward(ctx, pos, runner, *params):
        energy, forces = runner.values()
        ctx.runner = runner
        ctx.n_params = len(params)
        ws = runner.ws
        direct = ws.dfh_out.clone() if ws.dfh_idx is not None else energy.new_zeros(0)
        return energy.clone(), forces.clone(), direct      # (fresh tensors: autograd attaches this node to what forward returns)

    @staticmethod
    def backward(ctx, g_energy, g_forces, g_direct):
        r = ctx.runner
        ws = r.ws
        if g_energy is None:
            g_energy = torch.zeros(ws.B, dtype=torch.float32, device=ws.energy.device)
        r.grads(g_energy, g_forces, g_direct if ws.dfh_idx is not None else None)
        ws.busy = False
        # one copy of the flat gradient; autograd receives views of it (AccumulateGrad keeps them as the .grad tensors), and
        # distributed.allreduce_gradients recognises the flat layout and reduces it in place
        flat = ws.flat_grad.clone()
        out, off = [], 0
        for p in ws.params:
            out.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return (None, None) + tuple(out)


def acquire_workspace(model, g: hip.Graph, device, static: bool = False) -> TrainWorkspace:
    """A free workspace of the model's cache that fits the batch behind `g` (same N and B, at least its edge count, same
    parameter objects), or a new one."""
    cache = model.__dict__.setdefault('_train_ws', [])
    params = trainable_parameters(model)
    ws = next((w for w in cache if not w.busy and (w.N, w.B, w.energy.device) == (g.n_atoms, g.n_mol, device)
               and g.n_edges <= w.E and all(a is b for a, b in zip(w.params, params))), None)
    if ws is None:
        # capacity: real batches differ in their edge count from step to step -- leave 12.5 % head room (even, >= 64) so that
        # the next batches of the same shape reuse the buffers and the uploaded problem tables
        e_cap = g.n_edges if static else ((g.n_edges + g.n_edges // 8 + 64) & ~1)
        ws = TrainWorkspace(model, g.n_atoms, e_cap, g.n_mol, device)
        cache[:] = [w for w in cache if w.busy or (w.N, w.B) != (ws.N, ws.B)][-3:]   # superseded capacities go, a few shapes stay
        cache.append(ws)
    return ws


def forward_train(model, z, pos, cell, batch, graph: Optional[hip.Graph] = None):
    """Train-mode energy + gradient_force through the fused node.  `graph`: a static candidate list (GraphedTrainStep) whose
    geometry is refreshed at `pos`; otherwise the exact list is built (one host sync for the edge count).
    Returns (energy, gradient force, direct force or None, graph, workspace).  The outputs live in the workspace: valid until the next forward that reuses it (a
    workspace is reused only after its backward ran)."""
    emb = model.embedding_layers.edge_embedding
    zc = z.contiguous() if z.dtype == torch.int64 else z.long().contiguous()
    bc = batch.contiguous() if batch.dtype == torch.int64 else batch.long().contiguous()
    pd, cd = hip._f32c(pos.detach(), 'pos'), hip._f32c(cell.detach(), 'cell')
    with torch.no_grad():
        if graph is not None:
            g = hip.refresh_graph(graph, pd, cd, bc, emb.cutoff, emb.embedding.frequencies)
        else:
            g = hip.build_graph(pd, cd, bc, emb.cutoff, emb.embedding.frequencies, want_rbf=True, z=zc, envelope=emb.envelope_id)
    ws = acquire_workspace(model, g, pos.device, static=graph is not None)
    ws.busy = torch.is_grad_enabled()
    runner = Runner(model, zc, pd, cd, bc, g, ws)
    energy, forces, direct = FusedEnergyForces.apply(pos, runner, *ws.params)
    return energy, forces, (direct if ws.dfh_idx is not None else None), g, ws
