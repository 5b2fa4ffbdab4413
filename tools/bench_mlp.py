#!/usr/bin/env python3
"""Correctness + micro-benchmark of the fused edge MLP kernel (nnhip_mlp128) against torch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newtonnet_amd import hip

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

torch.manual_seed(0)
for M in [int(a) for a in sys.argv[1:]] or [1000, 21504, 313006]:
    X = torch.randn(M, 128, device='cuda'); W1 = torch.randn(128, 128, device='cuda') / 11; W2 = torch.randn(128, 128, device='cuda') / 11
    H = torch.empty(M, 128, device='cuda'); Y = torch.empty(M, 128, device='cuda')
    hip.mlp128(X, W1, W2, H, Y, 0)
    Hr = (X.double() @ W1.double().T); Yr = torch.nn.functional.silu(Hr) @ W2.double().T
    e_h, e_y = (H.double() - Hr).abs().max().item(), (Y.double() - Yr).abs().max().item()
    rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()  # noqa: E731
    r_h, r_y = rel(H, Hr), rel(Y, Yr)
    # adjoint: G = (Xg W1^T) * silu'(H), Yb = G W2^T (+ old)
    Xg = torch.randn(M, 256, device='cuda')[:, :128]          # strided view (ld = 256)
    Hd = Hr.float().contiguous(); Yb = torch.randn(M, 128, device='cuda'); Y0 = Yb.clone()
    s = torch.sigmoid(Hr); ds = s * (1 + Hr * (1 - s))
    Gr = (Xg.double() @ W1.double().T) * ds; Ybr = Gr @ W2.double().T
    hip.mlp128(Xg, W1, W2, Hd, Yb, 1, accumulate=True)
    e_b = (Yb.double() - (Y0.double() + Ybr)).abs().max().item()
    hip.mlp128(Xg, W1, W2, Hd, Yb, 1, accumulate=False)
    e_b2 = (Yb.double() - Ybr).abs().max().item()
    fl = 4.0 * M * 128 * 128
    t_f = timeit(lambda: hip.mlp128(X, W1, W2, H, Y, 0))
    t_b = timeit(lambda: hip.mlp128(Xg, W1, W2, Hd, Yb, 1))
    t_a = timeit(lambda: hip.mlp128(Xg, W1, W2, Hd, Yb, 1, accumulate=True))
    t_2 = timeit(lambda: (hip.linear128(X, W1, H), hip.linear128(H, W2, Y, prologue=hip.PRO_SILU)))
    print(f'M={M:7d} rel-rms H {r_h:.2e} Y {r_y:.2e} bwd {rel(Yb, Ybr):.2e} | err H {e_h:.1e} Y {e_y:.1e} bwd+acc {e_b:.1e} bwd {e_b2:.1e} | fwd {t_f:7.1f}us {fl/t_f/1e6:6.1f}TF | '
          f'bwd {t_b:7.1f}us {fl/t_b/1e6:6.1f}TF | bwd+acc {t_a:7.1f}us | 2x lin128 {t_2:7.1f}us', flush=True)
