"""bf16 GEMM rate on square reference shapes and on the backbone's shapes (random operands, HIP events, interleaved rounds:
the guide's methodology rules 24-25).  TILE=<policy> selects a kernel (see ug_gemm_set_tile_policy)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
T = 12336
cases = [("sq4096", 4096, 4096, 4096, "fwd"), ("sq8192", 8192, 8192, 8192, "fwd"),
         ("qkv_f", T, 2048, 1536, "fwd"), ("o_f", T, 1536, 1536, "resid"), ("gu_f", T, 17920, 1536, "fwd"), ("down_f", T, 1536, 8960, "resid"),
         ("gu_d", T, 1536, 17920, "dgrad"), ("down_d", T, 8960, 1536, "dgrad"), ("qkv_d", T, 1536, 2048, "dgrad"),
         ("gu_w", 17920, 1536, T, "wgrad"), ("down_w", 1536, 8960, T, "wgrad"), ("qkv_w", 2048, 1536, T, "wgrad"), ("o_w", 1536, 1536, T, "wgrad"),
         ("head_f", 4096, 159872, 1536, "fwd"), ("head_d", 4096, 1536, 159872, "dgrad"), ("head_w", 159872, 1536, 4096, "wgrad")]
only = os.environ.get("ONLY")
if only:
    cases = [c for c in cases if c[0] in only.split(",")]
pols = [int(p) for p in os.environ.get("TILES", "-1").split(",")]
rounds = int(os.environ.get("ROUNDS", "3"))
reps = int(os.environ.get("REPS", "4"))
for name, M, N, K, mode in cases:
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    if mode in ("fwd", "resid"):
        a, b = rnd(M, K), rnd(N, K)
        kw = dict()
        if mode == "resid":
            res = torch.randn(M, N, device=dev)
            kw = dict(epilogue=ops.UG_EPI_RESID, resid=res)
        run = lambda: ops.gemm(a, b, **kw)
    elif mode == "dgrad":
        a, b = rnd(M, K), rnd(K, N)
        run = lambda: ops.gemm(a, b, b_kmajor=True)
    else:
        a, b = rnd(K, M), rnd(K, N)
        out = torch.zeros(M, N, device=dev)
        run = lambda: ops.gemm(a, b, out=out, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
    best = {p: 0.0 for p in pols}
    for _ in range(rounds):
        for p in pols:
            ops.set_gemm_tile_policy(p)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            best[p] = max(best[p], 2.0 * M * N * K * reps / e0.elapsed_time(e1) / 1e9)
    print(f"{name:8s} M={M:6d} N={N:6d} K={K:6d} {mode:5s} " + "  ".join(f"pol{p}: {v:7.1f}" for p, v in best.items()), flush=True)
ops.set_gemm_tile_policy(-1)
