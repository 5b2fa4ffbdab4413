"""Train-mode forward of the hot path: differentiable twice (force-loss training).

The reference trains by back-propagating through the autograd force (`create_graph=True`, output.py:66-73;
trainer.py:307-309).  Here the graph build, the scatter-sums and the row gathers are HIP kernels wrapped as
autograd Functions that are linear maps and each other's adjoints (nnhip_segment_sum / nnhip_gather_rows), so
torch.autograd can differentiate the backward pass again; the dense linears go to the vendor GEMM through
torch.nn.functional.linear (plain library GEMMs) and the remaining elementwise algebra is torch on the GPU.
The fused inference kernels (csrc/pipeline.hip) are not used in train mode.

Mirrors: EmbeddingNet.forward newtonnet.py:139-161, EdgeEmbedding.forward representations.py:20-43,
InteractionNet.forward newtonnet.py:207-237, EnergyOutput/ScaleShift/EnergyAggregator output.py:98-100,
scalers.py:55-58, output.py:246.
"""
from __future__ import annotations

import torch
import torch.nn.functional as Fn

from newtonnet_amd import hip


class _EdgeGraph:
    """Index arrays of one batch for the Functions below (int32 on the device)."""
    def __init__(self, g: hip.Graph):
        self.row_ptr, self.col, self.rev = g.row_ptr, g.col, g.rev
        self.row = g.edge_index[0].to(torch.int32)          # receiver of each edge
        self.n_atoms, self.n_edges = g.n_atoms, g.n_edges


class SegmentSum(torch.autograd.Function):
    """out[i] = sum_{e in row i} x[e]  (scatter_sum over the receiver, newtonnet.py:214,226)."""
    @staticmethod
    def forward(ctx, x, eg):
        ctx.eg = eg
        return hip.segment_sum(x, eg.row_ptr, eg.n_atoms)

    @staticmethod
    def backward(ctx, gy):
        return Gather.apply(gy.contiguous(), ctx.eg, 'row'), None


class Gather(torch.autograd.Function):
    """out[e] = x[idx[e]] with idx = receiver ('row'), sender ('col') or reverse edge ('rev')."""
    @staticmethod
    def forward(ctx, x, eg, which):
        ctx.eg, ctx.which = eg, which
        return hip.gather_rows(x, getattr(eg, which))

    @staticmethod
    def backward(ctx, gy):
        eg, which = ctx.eg, ctx.which
        gy = gy.contiguous()
        if which == 'row':
            return SegmentSum.apply(gy, eg), None, None
        if which == 'col':   # scatter over the sender == segment-sum of the reverse-edge permutation
            return SegmentSum.apply(Gather.apply(gy, eg, 'rev'), eg), None, None
        return Gather.apply(gy, eg, 'rev'), None, None      # rev is an involution


def _envelope(x, p=9):
    """PolynomialCutoff(p) (representations.py:155-171) for p > 0, CosineCutoff (:177-203) for p == -1: the selector of the
    C ABI (EdgeEmbedding.envelope_id), so this path trains against the function the inference kernels evaluate."""
    if p == -1:
        return 0.5 * (1.0 + torch.cos(torch.pi * x))
    return 1.0 - 0.5 * (p + 1) * (p + 2) * x.pow(p) + p * (p + 2) * x.pow(p + 1) - 0.5 * p * (p + 1) * x.pow(p + 2)


def _mlp(seq, x):
    """Linear -> activation -> Linear; seq[1] is the model's activation module (SiLU by default)."""
    return _lin(seq[1](_lin(x, seq[0].weight, seq[0].bias)), seq[2].weight, seq[2].bias)


class _Fp32(torch.autograd.Function):
    """Identity that pins a tensor (and its gradient) to float32 at the boundary of a bf16 autocast region."""
    @staticmethod
    def forward(ctx, x):
        return x.float()

    @staticmethod
    def backward(ctx, g):
        return g.float()


def _lin(x, weight, bias=None):
    """Dense linear.  Under `torch.autocast('cuda', torch.bfloat16)` (BASELINE configs[2]: bf16 training step) the vendor
    GEMM runs with bf16 operands and fp32 accumulation; everything around it (gathers, segment sums, products,
    the radial basis) stays fp32."""
    y = Fn.linear(x, weight, bias)
    return y.float() if y.dtype != torch.float32 else y


def forward_train(model, z, pos, cell, batch, energy_idx: int, graph=None):
    """Returns (energy [B], atom_node, force_node, edge_index, graph); everything attached to autograd.

    graph: a STATIC candidate list (hip.build_graph with a cutoff of at least the model's, typically all pairs of each
    molecule) reused across steps -- no neighbor-list build and no host sync in the step, which makes the whole training
    step capturable in a HIP graph (distributed.GraphedTrainStep).  Its geometry is re-evaluated at `pos` and candidates at or
    beyond the cutoff are masked to exactly zero (the envelope and its first two derivatives vanish there)."""
    emb = model.embedding_layers
    ee = emb.edge_embedding
    static = graph is not None
    with torch.no_grad():
        if static:
            g = hip.refresh_graph(graph, pos.detach(), cell.detach(), batch, ee.cutoff, ee.embedding.frequencies)
            eg = getattr(g, '_train_eg', None)
            if eg is None:
                eg = g._train_eg = _EdgeGraph(g)
        else:
            g = hip.build_graph(pos.detach(), cell.detach(), batch, ee.cutoff, ee.embedding.frequencies, envelope=ee.envelope_id)
            eg = _EdgeGraph(g)
    i, j = g.edge_index[0], g.edge_index[1]
    # disp = pos_i - pos_j - (constant periodic image shift found by the neighbor kernel)
    shift = (pos.detach()[i] - pos.detach()[j]) - g.disp
    disp = pos[i] - pos[j] - shift
    r = disp.norm(dim=-1, keepdim=True)
    u = disp / r
    x = r / ee.cutoff
    rbf = _envelope(x, ee.envelope_id) * (torch.sin(ee.embedding.frequencies * x) / x)
    inside = None
    if static:   # candidates outside the cutoff: rbf -> 0 AND phi -> 0 (phi = W2 act(0) vanishes by itself only if act(0) = 0)
        inside = (r.detach().float() < ee.cutoff).to(rbf.dtype)
        rbf = rbf * inside

    a = emb.node_embedding(z)
    f = torch.zeros(z.shape[0], 3, emb.n_features, dtype=pos.dtype, device=pos.device)
    for l, il in enumerate(model.interaction_layers):
        m = _mlp(il.message_nodepart, a)
        with torch.autocast('cuda', enabled=False):   # K = 20 radial filter: always fp32
            eps = Fn.linear(rbf.float(), il.message_edgepart.weight.float())
        msg = eps * Gather.apply(m, eg, 'row') * Gather.apply(m, eg, 'col')
        a = a + SegmentSum.apply(msg, eg)
        phi1 = _mlp(il.equiv_message1, msg)
        if inside is not None:
            phi1 = phi1 * inside
        eq = phi1.unsqueeze(1) * u.unsqueeze(2)
        if l > 0:   # force_node == 0 entering the first layer (newtonnet.py:143)
            phi2 = _mlp(il.equiv_message2, msg)
            if inside is not None:
                phi2 = phi2 * inside
            eq = eq + phi2.unsqueeze(1) * Gather.apply(f, eg, 'col')
        f = f + SegmentSum.apply(eq.contiguous(), eg)
        a = a + (f * _lin(f, il.equiv_update.weight)).sum(dim=1)
        if il.layer_norm is not None:          # newtonnet.py:228-231
            a = il.layer_norm(a)

    head = model.output_layers[energy_idx].layers
    e = _lin(head[3](_lin(head[1](_lin(a, head[0].weight, head[0].bias)), head[2].weight, head[2].bias)),
             head[4].weight, head[4].bias)
    sc = model.scalers[energy_idx]
    if sc.scale is not None:
        e = e * sc.scale(z)
    if sc.shift is not None:
        e = e + sc.shift(z)
    energy = torch.zeros(cell.shape[0], dtype=e.dtype, device=e.device).index_add_(0, batch, e.reshape(-1))
    return energy, a, f, g
