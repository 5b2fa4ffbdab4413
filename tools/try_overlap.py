"""Experiment: does an HBM-bound row kernel (nnhip_segment_sum over [E,128]) overlap with the MFMA-bound persistent MLP kernel
(nnhip_mlp128) when they are launched on two streams?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newtonnet_amd import hip
M, E, N = 156503, 313006, 21504
X = torch.randn(M, 128, device='cuda'); W1 = torch.randn(128, 128, device='cuda') / 11; W2 = torch.randn(128, 128, device='cuda') / 11
H = torch.empty(M, 128, device='cuda'); Y = torch.empty(M, 128, device='cuda')
xe = torch.randn(E, 128, device='cuda')
row_ptr = torch.arange(0, E + 1, E // N, device='cuda', dtype=torch.int32)[:N + 1].contiguous()
row_ptr[-1] = E
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def mlp(n=4):
    for _ in range(n): hip.mlp128(X, W1, W2, H, Y, 0)
ye = torch.empty_like(xe)
def seg(n=6):     # HBM-bound stand-in: 160 MB read + 160 MB write per call
    for _ in range(n): torch.add(xe, 1.0, out=ye)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(3):
    with torch.cuda.stream(sa): mlp()
    with torch.cuda.stream(sb): seg()
torch.cuda.synchronize()
def only_a():
    with torch.cuda.stream(sa): mlp()
def only_b():
    with torch.cuda.stream(sb): seg()
def both():
    with torch.cuda.stream(sa): mlp()
    with torch.cuda.stream(sb): seg()
for _ in range(2):
    ta, tb, tab = timed(only_a), timed(only_b), timed(both)
    print(f'mlp x4 alone {ta:.3f} ms | add x6 alone {tb:.3f} ms | both streams {tab:.3f} ms (sum {ta+tb:.3f}, max {max(ta,tb):.3f})')
