"""Weight-gradient GEMM dW[N_out, K_in] = dY^T X on the backbone's shapes in the operand-layout forms the kernel takes
(k = token-major as the activations lie in memory, r = a transposed copy with the tokens contiguous), to price what a
producer-side transposed copy of an operand buys.  Interleaved rounds, HIP events.
Measured (round 2, TF/s): gate_up kk 931 / rk 920 / rr 753, down 972 / 1043 / 807, qkv 795 / 857 / 780, o 715 / 741 / 702 --
the transposing LDS reads are NOT what holds the weight gradients back (a token-contiguous operand is fetched in 64-byte
pieces of 256 rows 24 KB apart and loses more than the plain reads win); their 420- / 210-tile grids fill 82 % of the
rounds they occupy."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
T = int(os.environ.get("T", "12336"))
cases = [("gate_up", 17920, 1536), ("down", 1536, 8960), ("qkv", 2048, 1536), ("o", 1536, 1536)]
rounds, reps = int(os.environ.get("ROUNDS", "3")), int(os.environ.get("REPS", "4"))
for name, M, N in cases:
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    dy, x = rnd(T, M), rnd(T, N)
    dyT, xT = dy.t().contiguous(), x.t().contiguous()
    out = torch.zeros(M, N, device=dev)
    forms = {"kk": lambda: ops.gemm(dy, x, out=out, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1),
             "rk": lambda: ops.gemm(dyT, x, out=out, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1),
             "rr": lambda: ops.gemm(dyT, xT, out=out, epilogue=ops.UG_EPI_F32, beta=1)}
    ref = None
    best = {k: 0.0 for k in forms}
    for _ in range(rounds):
        for k, run in forms.items():
            out.zero_(); run(); torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                assert torch.allclose(out, ref, rtol=1e-3, atol=1e-2), (name, k, (out - ref).abs().max().item())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            best[k] = max(best[k], 2.0 * M * N * T * reps / e0.elapsed_time(e1) / 1e9)
    us = {k: 2.0 * M * N * T / v / 1e6 for k, v in best.items()}
    print(f"{name:8s} dW[{M},{N}] T={T}: " + "  ".join(f"{k}: {v:7.1f} TF/s ({us[k]:6.1f} us)" for k, v in best.items()), flush=True)
