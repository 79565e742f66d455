"""Micro-benchmark of the bf16 GEMM's three layout modes on the backbone's shapes (HIP events)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
M = 12336
shapes = {"qkv": (2048, 1536), "o": (1536, 1536), "gu": (17920, 1536), "down": (1536, 8960)}
reps = int(os.environ.get("REPS", "5"))
pol = int(os.environ.get("TILE", "-1"))
ops.set_gemm_tile_policy(pol)
print("tile policy", pol)
for name, (N, K) in shapes.items():
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    gw = torch.zeros(N, K, device=dev)
    res = {}
    for mode in ("fwd", "dgrad", "wgrad"):
        def run():
            if mode == "fwd":
                ops.gemm(x, w)
            elif mode == "dgrad":
                ops.gemm(dy, w, b_kmajor=True)
            else:
                ops.gemm(dy, x, out=gw, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[mode] = 2.0 * M * N * K / ms / 1e9
    print(f"{name:5s} M={M} N={N} K={K}: " + "  ".join(f"{k} {v:7.1f} TF/s" for k, v in res.items()), flush=True)
