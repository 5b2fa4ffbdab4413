#!/usr/bin/env python
"""Microbenchmark of the weight-gradient launch (csrc/train.hip: wgrad_kernel / wgrad_bf16_kernel + wgrad_reduce_kernel) on
pair-level problems of the config-2 size (P = 156 503 pair rows), by operand form.  Prints time, executed TFLOP/s and the
operand bytes per second.  usage: python tools/bench_wgrad.py [P] [chunks]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newtonnet_amd import hip  # noqa: E402

F = 128


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 156503
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device('cuda')
    L = hip.lib()
    g = torch.Generator(device='cuda').manual_seed(0)
    bufs = [torch.randn(P, F, device=dev, generator=g) for _ in range(6)]
    wide = torch.randn(P, 2 * F, device=dev, generator=g)
    out = torch.zeros(16, F, F, device=dev)
    p = lambda t, off=0: t.data_ptr() + 4 * off  # noqa: E731

    def prob(kind, k):
        q = hip.WgradProblem()
        q.M, q.activation, q.out = -1, 0, p(out[k])
        if kind == 'plain2':
            q.A1, q.B1, q.A2, q.B2, q.type = p(bufs[0]), p(bufs[1]), p(bufs[2]), p(bufs[3]), hip.WG_PLAIN
        elif kind == 'plain1':
            q.A1, q.B1, q.type = p(bufs[0]), p(bufs[1]), hip.WG_PLAIN
        elif kind == 'act':
            q.A1, q.A2, q.hB, q.dhB, q.type, q.lda1, q.lda2 = p(wide), p(wide, F), p(bufs[0]), p(bufs[1]), hip.WG_ACT, 2 * F, 2 * F
        elif kind == 'tdact':
            q.A1, q.B1, q.A2, q.B2, q.hA, q.type = p(bufs[0]), p(bufs[1]), p(bufs[2]), p(bufs[3]), p(bufs[4]), hip.WG_TDACT
        return q

    arrays = {'plain1': 2, 'plain2': 4, 'act': 4, 'tdact': 5}
    products = {'plain1': 1, 'plain2': 2, 'act': 2, 'tdact': 2}
    for bf16 in (0, 1):
        for kind, n in (('plain1', 10), ('plain2', 10), ('act', 10), ('tdact', 10), ('mix', 10)):
            kinds = [('act', 'tdact')[k & 1] for k in range(n)] if kind == 'mix' else [kind] * n
            tab = (hip.WgradProblem * n)(*[prob(kd, k) for k, kd in enumerate(kinds)])
            tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
            slabs = torch.empty(L.nnhip_wgrad_slab_bytes(n, chunks) // 4, dtype=torch.float32, device=dev)
            st = hip._stream(dev)

            def run():
                hip._check(L.nnhip_wgrad_batch(tab_dev.data_ptr(), n, chunks, slabs.data_ptr(), bf16, P, st), 'wgrad')
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            flop = sum(products[kd] for kd in kinds) * 2.0 * P * F * F
            byts = sum(arrays[kd] for kd in kinds) * P * F * 4.0
            print(f'{"bf16" if bf16 else "fp32"} {kind:7s} x{n}: {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s  '
                  f'{byts / ms / 1e9:6.2f} TB/s operands  (P={P}, chunks={chunks})', flush=True)


def real():
    """The weight-gradient launch of a real aspirin-1024 training step (the workspace's own problem table and buffers), whole
    and by problem class."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from bench_train import aspirin_batch
    from newtonnet_amd.distributed import FusedClipAdam, TrainStep
    from newtonnet_amd.models import NewtonNet
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    args = [t.cuda() for t in aspirin_batch(B)]
    torch.manual_seed(0)
    model = NewtonNet(output_properties=['energy', 'gradient_force']).to('cuda')
    model.train()
    step = TrainStep(model, FusedClipAdam(model, lr=1e-3, max_norm=1.0), 1.0, 50.0)
    for _ in range(2):
        step(*args)
    torch.cuda.synchronize()
    ws = model._train_ws[-1]
    L, dev = hip.lib(), torch.device('cuda')
    n = ws.n_probs
    host = (hip.WgradProblem * n).from_buffer_copy(ws.prob_dev.cpu().numpy().tobytes())
    Pn = int(args[1].shape[0])
    E = None
    for w in model._train_ws:
        E = w.c.n_edges
    P = E // 2
    print(f'N={Pn} P={P} problems={n} chunks={ws.chunks}')
    classes = {'all': list(range(n)),
               'pair128': [k for k in range(n) if host[k].M < 0 and not host[k].b_cols32],
               'pair_nb32': [k for k in range(n) if host[k].M < 0 and host[k].b_cols32],
               'node': [k for k in range(n) if host[k].M >= 0]}
    for bf16 in (0, 1):
        for name, idx in classes.items():
            if not idx:
                continue
            tab = (hip.WgradProblem * len(idx))(*[host[k] for k in idx])
            tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
            st = hip._stream(dev)

            def run():
                hip._check(L.nnhip_wgrad_batch(tab_dev.data_ptr(), len(idx), ws.chunks, ws.slabs.data_ptr(), bf16, P, st), 'wgrad')
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            flop = sum((2 if host[k].A2 else 1) * 2.0 * (P if host[k].M < 0 else host[k].M) * F * (32 if host[k].b_cols32 else F)
                       for k in idx)
            print(f'{"bf16" if bf16 else "fp32"} {name:10s} ({len(idx):2d} problems): {ms:7.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s',
                  flush=True)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'real':
    real()
    sys.exit(0)
if __name__ == '__main__':
    main()
