"""Per-tile fixed cost of the multi-round GEMM launches: the same M x N at K and 2K (time = rounds x (fixed + per-k x K))."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
M = 12336


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, N in (("gate_up forward + SwiGLU", 17920), ("plain forward N=17920", 17920), ("plain forward N=1536 (one round)", 1536)):
    res = []
    for K in (1536, 3072, 4608):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        fn = (lambda: ops.gemm_swiglu(x, w)) if "SwiGLU" in name else (lambda: ops.gemm(x, w))
        res.append((K, timed(fn)))
    (k1, t1), (k2, t2), (k3, t3) = res
    per_k = (t3 - t1) / (k3 - k1)
    fixed = t1 - per_k * k1
    print(f"{name}: " + "  ".join(f"K={k}: {t:7.1f} us" for k, t in res) + f"   -> fixed {fixed:6.1f} us per launch = {100 * fixed / t1:4.1f} % at K=1536", flush=True)
