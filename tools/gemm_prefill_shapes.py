"""The residual-epilogue GEMMs of the AR prefill (M = 16 rows x 138 tokens = 2 208): o projection (N = K = 1536) and down projection
(N = 1536, K = 8960) under the automatic selection and under forced forms (policy 8: every tile cut along K into private fp32
partials; 3: plain 256 x 256 tiles; 0 / 2: 128 x 128).  TF/s, 20 launches each, random operands."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
for M in (2208, 16 * 387, 4416):
    for N, K in ((1536, 1536), (1536, 8960), (2048, 1536)):
        g = torch.Generator(device=dev).manual_seed(1)
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        res = torch.randn(M, N, device=dev, generator=g)
        out = torch.empty(M, N, device=dev)
        line = [f"M={M} N={N} K={K}:"]
        for pol in (-1, 8, 3, 0):
            ops.set_gemm_tile_policy(pol)
            try:
                fn = lambda: ops.gemm(a, b, out=out, epilogue=ops.UG_EPI_RESID, resid=res)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                line.append(f"policy {pol}: {us:6.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF/s |")
            except Exception as e:
                line.append(f"policy {pol}: {type(e).__name__} |")
            finally:
                ops.set_gemm_tile_policy(-1)
        print(" ".join(line), flush=True)
