"""A few launches of one backbone GEMM (gate_up forward, M = 12336) for rocprofv3 --pmc passes (HBM traffic per launch)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
M, N, K = 12336, 17920, 1536
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.gemm(x, w, out=out)
torch.cuda.synchronize()
