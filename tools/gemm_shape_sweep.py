"""The layer's eight forward / dgrad GEMM launches at a given token count M under every tile policy: which kernel wins where
(the selection rule of gemm_bf16.hip::launch is checked against this table).  usage: gemm_shape_sweep.py M [policies...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
M = int(sys.argv[1])
pols = [int(p) for p in sys.argv[2:]] or [-1, 0, 3, 10, 44, 45, 46, 47, 49, 50, 51]
reps = int(os.environ.get("REPS", "10"))
shapes = [("qkv fwd", 2048, 1536, False), ("o fwd", 1536, 1536, False), ("gu fwd", 17920, 1536, False), ("down fwd", 1536, 8960, False),
          ("qkv dgrad", 1536, 2048, True), ("o dgrad", 1536, 1536, True), ("gu dgrad", 1536, 17920, True), ("down dgrad", 8960, 1536, True)]
print(f"M={M}  us per launch (TF/s)   policies: {pols}")
for name, N, K, bk in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(K, N, device=dev) if bk else torch.randn(N, K, device=dev)).mul_(0.02).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = []
    for pol in pols:                                   # one untimed pass over every policy: the first timed one is not the cold one
        ops.set_gemm_tile_policy(pol)
        for _ in range(3):
            ops.gemm(a, b, out=out, b_kmajor=bk)
    for pol in pols:
        ops.set_gemm_tile_policy(pol)
        for _ in range(2):
            ops.gemm(a, b, out=out, b_kmajor=bk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm(a, b, out=out, b_kmajor=bk)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        row.append(f"{pol}:{us:6.1f}")
    ops.set_gemm_tile_policy(-1)
    print(f"{name:10s} N={N:5d} K={K:5d}  " + "  ".join(row), flush=True)
