#!/usr/bin/env python3
"""Micro-benchmark of the dense 128x128 linear kernels through the C ABI (nnhip_linear128).
usage: python tools/bench_lin.py [M ...]      env: NNHIP_LIN_BLOCKS, NNHIP_SMALL_TILES"""
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

Ms = [int(a) for a in sys.argv[1:]] or [21504, 64512, 313006]
W = torch.randn(128, 128, device='cuda') / 11
b = torch.randn(128, device='cuda')
print('NNHIP_LIN_BLOCKS', os.environ.get('NNHIP_LIN_BLOCKS'), 'NNHIP_SMALL_TILES', os.environ.get('NNHIP_SMALL_TILES'))
for M in Ms:
    A = torch.randn(M, 128, device='cuda'); H = torch.randn(M, 128, device='cuda'); C = torch.empty_like(A)
    ref = A @ W.T
    out = hip.linear128(A, W, C)
    err = (out - ref).abs().max().item()
    fl = 2.0 * M * 128 * 128
    res = []
    for name, kw in [('store', {}), ('bias', dict(bias=b, epilogue=hip.EPI_BIAS)), ('silu>', dict(prologue=hip.PRO_SILU)),
                     ('dsilu', dict(H=H, epilogue=hip.EPI_DSILU)), ('acc', dict(epilogue=hip.EPI_ACC))]:
        us = timeit(lambda: hip.linear128(A, W, C, **kw))
        res.append(f'{name} {us:7.1f}us {fl / us / 1e6:6.1f}TF')
    us = timeit(lambda: torch.mm(A, W.T, out=C))
    res.append(f'torch.mm {us:7.1f}us {fl / us / 1e6:6.1f}TF')
    print(f'M={M:8d} err {err:.1e} | ' + ' | '.join(res), flush=True)
