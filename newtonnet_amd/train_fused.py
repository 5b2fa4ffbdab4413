"""Train-mode forward of the hot path on the hand-written kernels: (theta, pos) -> (energy, forces) as ONE autograd node.

The reference trains by back-propagating through the autograd force (`create_graph=True`, newtonnet/models/output.py:66-73;
newtonnet/train/trainer.py:299-313; loss newtonnet/train/loss.py:48,72,96).  Here nothing records a graph: `FusedEnergyForces`
runs the value sweeps (forward + analytic force adjoint) stage by stage through the C ABI with every intermediate kept, and its
backward -- given dL/dE and dL/dF of ANY loss -- returns the parameter gradients as "tangent over reverse":
    dL/dtheta = sum_b c_b dE_b/dtheta - D_d[grad_theta E_tot],      c = dL/dE,  d = dL/dF,
the reverse sweep differentiated once more in forward mode along v = -d (csrc/train.hip), followed by one batched launch of
split-K MFMA weight-gradient products.  Every kernel is an entry point of include/newtonnet_hip.h ("Per-stage entry points");
this file only owns buffers and order.  tests/tangent_ref.py states the same four sweeps in fp64 torch.

Supported: output_properties {'energy', 'gradient_force'}, layer_norm=False, every fused activation.  Other head sets
(energy only, direct_force, layer_norm=True) keep the torch-graph path of train_ops.py.  dL/dpos is not produced (None): the
reference would return it, no trainer uses it.
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
    return (sorted(keys) == ['energy', 'gradient_force'] and all(il.layer_norm is None for il in model.interaction_layers)
            and model.embedding_layers.n_features == F)


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
        self._tables(model, device)

    # device-resident tables of the batched launches (pointer lists of the transposes / filter tables stay on the host)
    def _tables(self, model, device):
        L, N, P = self.L, self.N, self.P
        act = hip.ACTIVATION_IDS[model.activation_name]
        gmap = {id(p): g for p, g in zip(self.params, self.grads)}
        G = lambda prm: gmap[id(prm)]  # noqa: E731
        probs, sums = [], []

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
        head = model.output_layers[list(model.output_properties).index('energy')].layers
        add(G(head[2].weight), N, self.dg_e2, A2=self.g_e2, typ=hip.WG_ACT, hB=self.e1, dhB=self.de1)
        add(G(head[0].weight), N, self.dg_e1, B1=self.a_out[L - 1], A2=self.t_e1, B2=self.da_out[L - 1], typ=hip.WG_TDACT,
            hA=self.e1)
        colsum(G(head[2].bias), self.dg_e2, N)
        colsum(G(head[0].bias), self.dg_e1, N)
        colsum(G(head[4].weight), self.w4row, N)
        self.g_head4_b = G(head[4].bias)
        self.n_probs, self.n_sums = len(probs), len(sums)
        arr = (hip.WgradProblem * len(probs))(*probs)
        self.prob_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        arr = (hip.ColsumProblem * len(sums))(*sums)
        self.sum_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.cs_scratch = torch.empty(max(hip.lib().nnhip_colsum_scratch_bytes(self.n_sums) // 4, 1), dtype=torch.float32,
                                      device=device)
        self.chunks = max(1, min(256, (max(P, N) + 31) // 32))
        self.slabs = torch.empty(hip.lib().nnhip_wgrad_slab_bytes(self.n_probs, self.chunks) // 4, dtype=torch.float32,
                                 device=device)


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

    # -- small helpers ------------------------------------------------------------------------------------------------
    def _desc(self, mode, X, W1, W2, H, Y, M, *, ldx=F, b1=None, b2=None, accumulate=False, T=None, T2=None, Hd=None, G=None,
              x_off=0):
        d = hip.MlpDesc()
        d.X, d.ldx = _p(X, x_off).value, ldx
        d.W1, d.W2 = _p(W1).value, _p(W2).value
        d.b1, d.b2 = (_p(b1).value if b1 is not None else None), (_p(b2).value if b2 is not None else None)
        d.H, d.ldh, d.Y, d.ldy = _p(H).value, F, _p(Y).value, F
        d.M, d.mode, d.accumulate, d.activation = M, mode, 1 if accumulate else 0, self.act
        d.T = _p(T).value if T is not None else None
        d.T2 = _p(T2).value if T2 is not None else None
        d.Hd = _p(Hd).value if Hd is not None else None
        d.G = _p(G).value if G is not None else None
        return d

    def _mlp(self, mode, X, W1, W2, H, Y, M, **kw):
        if M > 0:
            _chk(hip.lib().nnhip_mlp128_ex(C.byref(self._desc(mode, X, W1, W2, H, Y, M, **kw)), self.st), 'nnhip_mlp128_ex')

    def _mlp2(self, d0, d1):
        """equiv_message1 | equiv_message2 (or their adjoints / tangents) over the same pair rows in one launch"""
        if d0.M > 0:
            _chk(hip.lib().nnhip_mlp128_pair_ex(C.byref(d0), C.byref(d1), self.st), 'nnhip_mlp128_pair_ex')

    def _lin(self, A, W, out, M, acc=False):
        if M > 0:
            _chk(hip.lib().nnhip_linear128(_p(A), F, _p(W), _p(out), F, None, None, 0, M, hip.PRO_NONE,
                                           hip.EPI_ACC if acc else hip.EPI_STORE, self.st), 'nnhip_linear128')

    def _prepare(self):
        """Parameter-only data of this step: transposed weights (one launch) and the radial-filter tables (one launch)."""
        L_, ws, model = hip.lib(), self.ws, self.model
        src, dst = [], []
        for l, il in enumerate(model.interaction_layers):
            for k, w in enumerate((il.message_nodepart[0].weight, il.message_nodepart[2].weight, il.equiv_message1[0].weight,
                                   il.equiv_message1[2].weight, il.equiv_message2[0].weight, il.equiv_message2[2].weight,
                                   il.equiv_update.weight)):
                src.append(w)
                dst.append(ws.wT[l][k])
        head = model.output_layers[self.energy_idx].layers
        src += [head[0].weight, head[2].weight]
        dst += ws.headT
        for o in range(0, len(src), 40):
            n = min(40, len(src) - o)
            a = (_vp * n)(*[t.data_ptr() for t in src[o:o + n]])
            b = (_vp * n)(*[t.data_ptr() for t in dst[o:o + n]])
            _chk(L_.nnhip_transpose128(a, b, n, self.st), 'nnhip_transpose128')
        n = ws.L
        ew = (_vp * n)(*[il.message_edgepart.weight.data_ptr() for il in model.interaction_layers])
        tb = (_vp * n)(*[t.data_ptr() for t in ws.ftab])
        emb = model.embedding_layers.edge_embedding
        _chk(L_.nnhip_filter_tables(ew, tb, n, _p(emb.embedding.frequencies), emb.n_basis, emb.envelope_id, self.st),
             'nnhip_filter_tables')

    # -- sweeps 1 and 2: values ------------------------------------------------------------------------------------------
    def values(self):
        L_, ws, g, model, st, act = hip.lib(), self.ws, self.g, self.model, self.st, self.act
        N, B, L = ws.N, ws.B, ws.L
        E, P = g.n_edges, g.n_edges // 2                     # this batch's own counts (the workspace holds capacities)
        self._prepare()
        layers = list(model.interaction_layers)
        head = model.output_layers[self.energy_idx].layers
        sc = model.scalers[self.energy_idx]
        z = self.z
        idx = (_p(g.row_ptr), _p(g.col), _p(g.pid))
        _chk(L_.nnhip_embed(_p(z), _p(model.embedding_layers.node_embedding.weight), N, _p(ws.a0), st), 'nnhip_embed')
        n0 = layers[0].message_nodepart
        self._mlp(hip.MODE_FWD, ws.a0, n0[0].weight, n0[2].weight, ws.hn[0], ws.m[0], N, b1=n0[0].bias, b2=n0[2].bias)
        a_in, f_in = ws.a0, None
        for l, il in enumerate(layers):
            _chk(L_.nnhip_message_fwd(_p(ws.m[l]), _p(g.xg), _p(ws.ftab[l]), *idx, _p(a_in), _p(ws.msg[l]), _p(ws.a_mid[l]), N, st),
                 'nnhip_message_fwd')
            e1w, e2w = il.equiv_message1, il.equiv_message2
            if l > 0:
                self._mlp2(self._desc(hip.MODE_FWD, ws.msg[l], e1w[0].weight, e1w[2].weight, ws.h1[l], ws.phi1[l], P),
                           self._desc(hip.MODE_FWD, ws.msg[l], e2w[0].weight, e2w[2].weight, ws.h2[l], ws.phi2[l], P))
            else:
                self._mlp(hip.MODE_FWD, ws.msg[l], e1w[0].weight, e1w[2].weight, ws.h1[l], ws.phi1[l], P)
            _chk(L_.nnhip_force_message_fwd(_p(ws.phi1[l]), _p(ws.phi2[l]), _p(g.geo), _p(g.xg), *idx, _p(f_in), _p(ws.f_out[l]),
                                            N, st), 'nnhip_force_message_fwd')
            if l + 1 < L:
                nx = layers[l + 1].message_nodepart
                nxt = (nx[0].weight, nx[0].bias, nx[2].weight, nx[2].bias, ws.hn[l + 1], ws.m[l + 1])
            else:
                nxt = (head[0].weight, head[0].bias, head[2].weight, head[2].bias, ws.e1, ws.e2)
            _chk(L_.nnhip_node_fwd(_p(ws.f_out[l]), _p(ws.a_mid[l]), _p(il.equiv_update.weight), _p(ws.q[l]), _p(ws.a_out[l]),
                                   *[_p(t) for t in nxt], N, act, st), 'nnhip_node_fwd')
            a_in, f_in = ws.a_out[l], ws.f_out[l]
        _chk(L_.nnhip_head_out(_p(ws.e2), _p(head[4].weight), _p(head[4].bias),
                               _p(sc.scale.weight) if sc.scale is not None else None,
                               _p(sc.shift.weight) if sc.shift is not None else None, _p(z), _p(g.mol_ptr), N, B, act,
                               _p(ws.atom_energy), _p(ws.g_e2), _p(ws.energy), st), 'nnhip_head_out')
        # ---- reverse (seed 1)
        self._mlp(hip.MODE_TAN, ws.g_e2, ws.headT[1], ws.headT[0], ws.e1, ws.GA[L - 1], N, T=ws.t_e1)
        _chk(L_.nnhip_node_bwd(None, None, None, None, _p(ws.GA[L - 1]), 0, _p(ws.f_out[L - 1]), _p(ws.q[L - 1]), None,
                               _p(ws.wT[L - 1][6]), _p(ws.gf[L - 1]), N, act, st), 'nnhip_node_bwd')
        pp = 0
        for l in range(L - 1, -1, -1):
            f_prev = ws.f_out[l - 1] if l > 0 else None
            Gf = ws.Gf[pp]
            _chk(L_.nnhip_force_message_bwd(_p(ws.gf[l]), _p(ws.phi1[l]), _p(ws.phi2[l]), _p(g.geo), _p(g.xg), *idx, _p(f_prev),
                                            _p(ws.g_h12[l]), _p(ws.g_u, 4 * l * E), _p(Gf), N, st), 'nnhip_force_message_bwd')
            d1 = self._desc(hip.MODE_TAN, ws.g_h12[l], ws.wT[l][3], ws.wT[l][2], ws.h1[l], ws.g_msg[l], P, ldx=2 * F, T=ws.t1[l])
            if l > 0:
                self._mlp2(d1, self._desc(hip.MODE_TAN, ws.g_h12[l], ws.wT[l][5], ws.wT[l][4], ws.h2[l], ws.g_msg[l], P, ldx=2 * F,
                                          T=ws.t2[l], accumulate=True, x_off=F))
            elif P > 0:
                _chk(L_.nnhip_mlp128_ex(C.byref(d1), st), 'nnhip_mlp128_ex')
            _chk(L_.nnhip_message_bwd(_p(ws.g_msg[l]), _p(ws.GA[l]), _p(ws.m[l]), _p(g.xg), _p(ws.ftab[l]), *idx,
                                      _p(ws.g_m[l]) if l > 0 else None, _p(ws.g_x, l * E), N, 1 if l > 0 else 0, st),
                 'nnhip_message_bwd')
            if l > 0:
                ws.GA[l - 1].copy_(ws.GA[l])
                self._mlp(hip.MODE_TAN, ws.g_m[l], ws.wT[l][1], ws.wT[l][0], ws.hn[l], ws.GA[l - 1], N, T=ws.t_n[l],
                          accumulate=True)
                _chk(L_.nnhip_node_bwd(None, None, None, None, _p(ws.GA[l - 1]), 0, _p(ws.f_out[l - 1]), _p(ws.q[l - 1]), _p(Gf),
                                       _p(ws.wT[l - 1][6]), _p(ws.gf[l - 1]), N, act, st), 'nnhip_node_bwd')
            pp ^= 1
        emb = model.embedding_layers.edge_embedding
        _chk(L_.nnhip_edge_embed_bwd(_p(ws.g_x), _p(ws.g_u), _p(g.geo), _p(g.disp), _p(self.pos), _p(self.cell), _p(g.row_ptr),
                                     _p(g.col), _p(g.rev), _p(g.mol_ptr), N, E, B, L, float(emb.cutoff), _p(ws.g_d),
                                     _p(ws.forces), None, st), 'nnhip_edge_embed_bwd')
        return ws.energy, ws.forces

    # -- sweeps 3 and 4: tangents, then the weight gradients ---------------------------------------------------------------
    def grads(self, g_energy: torch.Tensor, g_forces: torch.Tensor):
        L_, ws, g, model, st, act = hip.lib(), self.ws, self.g, self.model, self.st, self.act
        N, B, L = ws.N, ws.B, ws.L
        E, P = g.n_edges, g.n_edges // 2
        layers = list(model.interaction_layers)
        head = model.output_layers[self.energy_idx].layers
        sc = model.scalers[self.energy_idx]
        emb = model.embedding_layers.edge_embedding
        idx = (_p(g.row_ptr), _p(g.col), _p(g.pid))
        g_forces = g_forces.reshape(N, 3).to(torch.float32).contiguous()
        g_energy = g_energy.to(torch.float32).contiguous()
        # ---- sweep 3: tangent forward along v = -dL/dF
        _chk(L_.nnhip_edge_tangent_geom(_p(g_forces), -1.0, _p(g.edge_index), _p(g.geo), E, float(emb.cutoff), _p(ws.tgeo), st),
             'nnhip_edge_tangent_geom')
        for l, il in enumerate(layers):
            first = l == 0
            da_in = None if first else ws.da_out[l - 1]
            _chk(L_.nnhip_message_tan_fwd(_p(ws.m[l]), None if first else _p(ws.dm[l]), _p(g.xg), _p(ws.tgeo), _p(ws.ftab[l]), *idx,
                                          _p(da_in), _p(ws.dmsg[l]), _p(ws.da_mid), N, st), 'nnhip_message_tan_fwd')
            e1w, e2w = il.equiv_message1, il.equiv_message2
            d1 = self._desc(hip.MODE_TAN, ws.dmsg[l], e1w[0].weight, e1w[2].weight, ws.h1[l], ws.dphi1[l], P, T=ws.dh1[l])
            if not first:
                self._mlp2(d1, self._desc(hip.MODE_TAN, ws.dmsg[l], e2w[0].weight, e2w[2].weight, ws.h2[l], ws.dphi2[l], P,
                                          T=ws.dh2[l]))
            elif P > 0:
                _chk(L_.nnhip_mlp128_ex(C.byref(d1), st), 'nnhip_mlp128_ex')
            _chk(L_.nnhip_force_message_tan_fwd(_p(ws.phi1[l]), _p(ws.dphi1[l]), _p(ws.phi2[l]), _p(ws.dphi2[l]), _p(g.geo),
                                                _p(ws.tgeo), _p(g.xg), *idx, None if first else _p(ws.f_out[l - 1]),
                                                None if first else _p(ws.df_out[l - 1]), _p(ws.df_out[l]), N, st),
                 'nnhip_force_message_tan_fwd')
            self._lin(ws.df_out[l], il.equiv_update.weight, ws.dq[l], 3 * N)
            _chk(L_.nnhip_update_tan_fwd(_p(ws.da_mid), _p(ws.f_out[l]), _p(ws.df_out[l]), _p(ws.q[l]), _p(ws.dq[l]), N,
                                         _p(ws.da_out[l]), st), 'nnhip_update_tan_fwd')
            if l + 1 < L:
                nx = layers[l + 1].message_nodepart
                self._mlp(hip.MODE_TAN, ws.da_out[l], nx[0].weight, nx[2].weight, ws.hn[l + 1], ws.dm[l + 1], N, T=ws.dhn[l + 1])
            else:
                self._mlp(hip.MODE_TAN, ws.da_out[l], head[0].weight, head[2].weight, ws.e1, ws.de2, N, T=ws.de1)
        # ---- sweep 4: tangent reverse, seed tangent c = dL/dE
        _chk(L_.nnhip_head_seed_tan(_p(ws.e2), _p(ws.de2), _p(head[4].weight), _p(head[4].bias),
                                    _p(sc.scale.weight) if sc.scale is not None else None, _p(self.z), _p(self.batch),
                                    _p(g_energy), N, act, _p(ws.dg_e2), _p(ws.w4row), _p(ws.scal), st), 'nnhip_head_seed_tan')
        self._mlp(hip.MODE_TAN2, ws.dg_e2, ws.headT[1], ws.headT[0], ws.e1, ws.dGA, N, T2=ws.t_e1, Hd=ws.de1, G=ws.dg_e1)
        dGf, pp = None, 0
        for l in range(L - 1, -1, -1):
            first = l == 0
            _chk(L_.nnhip_update_tan_bwd(_p(ws.GA[l]), _p(ws.dGA), _p(ws.f_out[l]), _p(ws.df_out[l]), _p(ws.q[l]), _p(ws.dq[l]),
                                         _p(dGf), N, _p(ws.gq[l]), _p(ws.dgq[l]), _p(ws.dgf), st), 'nnhip_update_tan_bwd')
            self._lin(ws.dgq[l], ws.wT[l][6], ws.dgf, 3 * N, acc=True)
            nxt = ws.dGf[pp]
            _chk(L_.nnhip_force_message_tan_bwd(_p(ws.gf[l]), _p(ws.dgf), _p(ws.phi2[l]), _p(ws.dphi2[l]), _p(g.geo), _p(ws.tgeo),
                                                _p(g.xg), *idx, None if first else _p(ws.f_out[l - 1]),
                                                None if first else _p(ws.df_out[l - 1]), _p(ws.dg_h12[l]),
                                                None if first else _p(nxt), N, st), 'nnhip_force_message_tan_bwd')
            d1 = self._desc(hip.MODE_TAN2, ws.dg_h12[l], ws.wT[l][3], ws.wT[l][2], ws.h1[l], ws.dg_msg, P, ldx=2 * F, T2=ws.t1[l],
                            Hd=ws.dh1[l], G=ws.dg_h1[l])
            if not first:
                self._mlp2(d1, self._desc(hip.MODE_TAN2, ws.dg_h12[l], ws.wT[l][5], ws.wT[l][4], ws.h2[l], ws.dg_msg, P, ldx=2 * F,
                                          T2=ws.t2[l], Hd=ws.dh2[l], G=ws.dg_h2[l], accumulate=True, x_off=F))
            elif P > 0:
                _chk(L_.nnhip_mlp128_ex(C.byref(d1), st), 'nnhip_mlp128_ex')
            _chk(L_.nnhip_message_tan_bwd(_p(ws.g_msg[l]), _p(ws.dg_msg), _p(ws.GA[l]), _p(ws.dGA), _p(ws.m[l]),
                                          None if first else _p(ws.dm[l]), _p(g.xg), _p(ws.tgeo), _p(ws.ftab[l]), *idx,
                                          _p(ws.dg_m[l]), _p(ws.g_eps[l]), _p(ws.dg_eps[l]), N, st), 'nnhip_message_tan_bwd')
            self._mlp(hip.MODE_TAN2, ws.dg_m[l], ws.wT[l][1], ws.wT[l][0], ws.hn[l], ws.dGA, N,
                      T2=ws.zeros_nf if first else ws.t_n[l], Hd=ws.zeros_nf if first else ws.dhn[l], G=ws.dg_hn[l], accumulate=True)
            dGf = nxt
            pp ^= 1
        # ---- weight gradients: one batched split-K launch + its reduction, column sums, per-element sums
        _chk(L_.nnhip_pair_rbf(_p(g.rbf), _p(g.drbf), _p(ws.tgeo), _p(g.edge_index), _p(g.pid), E, emb.n_basis, _p(ws.rb), st),
             'nnhip_pair_rbf')
        _chk(L_.nnhip_wgrad_batch(_p(ws.prob_dev), ws.n_probs, ws.chunks, _p(ws.slabs), 1 if self.bf16 else 0, P, st),
             'nnhip_wgrad_batch')
        _chk(L_.nnhip_colsum_batch(_p(ws.sum_dev), ws.n_sums, _p(ws.cs_scratch), st), 'nnhip_colsum_batch')
        gmap = {id(p): gr for p, gr in zip(ws.params, ws.grads)}
        _chk(L_.nnhip_species_sum(_p(ws.dGA), F, F, _p(self.z), N, _p(ws.sp_scratch),
                                  _p(gmap[id(model.embedding_layers.node_embedding.weight)]), 0, F, F, None, 0, 0, 0, None, 0, st),
             'nnhip_species_sum')
        _chk(L_.nnhip_species_sum(_p(ws.scal), 4, 4, _p(self.z), N, _p(ws.sp_scratch),
                                  _p(gmap[id(sc.scale.weight)]) if sc.scale is not None else None, 0, 1, 1,
                                  _p(gmap[id(sc.shift.weight)]) if sc.shift is not None else None, 1, 1, 1,
                                  _p(ws.g_head4_b), 2, st), 'nnhip_species_sum')
        return ws.grads


class FusedEnergyForces(torch.autograd.Function):
    """(pos, *parameters) -> (energy [B], forces [N,3]); backward returns dL/dparameters (dL/dpos is not produced)."""

    @staticmethod
    def forward(ctx, pos, runner, *params):
        energy, forces = runner.values()
        ctx.runner = runner
        ctx.n_params = len(params)
        return energy.clone(), forces.clone()      # (fresh tensors: autograd attaches this node to what forward returns)

    @staticmethod
    def backward(ctx, g_energy, g_forces):
        r = ctx.runner
        ws = r.ws
        if g_energy is None:
            g_energy = torch.zeros(ws.B, dtype=torch.float32, device=ws.energy.device)
        if g_forces is None:
            g_forces = torch.zeros(ws.N, 3, dtype=torch.float32, device=ws.energy.device)
        r.grads(g_energy, g_forces)
        ws.busy = False
        # one copy of the flat gradient; autograd receives views of it (AccumulateGrad keeps them as the .grad tensors), and
        # distributed.allreduce_gradients recognises the flat layout and reduces it in place
        flat = ws.flat_grad.clone()
        out, off = [], 0
        for p in ws.params:
            out.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return (None, None) + tuple(out)


def forward_train(model, z, pos, cell, batch, graph: Optional[hip.Graph] = None):
    """Train-mode energy + gradient_force through the fused node.  `graph`: a static candidate list (GraphedTrainStep) whose
    geometry is refreshed at `pos`; otherwise the exact list is built (one host sync for the edge count).
    Returns (energy, forces, graph, workspace).  The outputs live in the workspace: valid until the next forward that reuses it (a
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
    cache = model.__dict__.setdefault('_train_ws', [])
    params = trainable_parameters(model)
    ws = next((w for w in cache if not w.busy and (w.N, w.B, w.energy.device) == (g.n_atoms, g.n_mol, pos.device)
               and g.n_edges <= w.E and all(a is b for a, b in zip(w.params, params))), None)
    if ws is None:
        # capacity: real batches differ in their edge count from step to step -- leave 12.5 % head room (even, >= 64) so that
        # the next batches of the same shape reuse the buffers and the uploaded problem tables
        e_cap = g.n_edges if graph is not None else ((g.n_edges + g.n_edges // 8 + 64) & ~1)
        ws = TrainWorkspace(model, g.n_atoms, e_cap, g.n_mol, pos.device)
        cache[:] = [w for w in cache if w.busy or (w.N, w.B) != (ws.N, ws.B)][-3:]   # superseded capacities go, a few shapes stay
        cache.append(ws)
    ws.busy = torch.is_grad_enabled()
    runner = Runner(model, zc, pd, cd, bc, g, ws)
    energy, forces = FusedEnergyForces.apply(pos, runner, *ws.params)
    return energy, forces, g, ws
