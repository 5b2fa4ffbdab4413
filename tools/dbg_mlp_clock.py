"""Tooling: run the MLP_CLOCK_DEBUG build of mlp128 (device printf of per-phase wall-clock stamps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newtonnet_amd import hip
M = int(sys.argv[1]) if len(sys.argv) > 1 else 156503
X = torch.randn(M, 128, device='cuda'); W1 = torch.randn(128, 128, device='cuda') / 11; W2 = torch.randn(128, 128, device='cuda') / 11
H = torch.empty(M, 128, device='cuda'); Y = torch.empty(M, 128, device='cuda')
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    hip.mlp128(X, W1, W2, H, Y, 0)
torch.cuda.synchronize()
