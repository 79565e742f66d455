"""down dgrad + SwiGLU backward at the benchmark shape: two launches against the fused epilogue (us per layer)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
M, I, K = int(os.environ.get("M", "12336")), 8960, 1536
dy = torch.randn(M, K, device=dev).to(torch.bfloat16)
wd = (torch.randn(K, I, device=dev) * 0.02).to(torch.bfloat16)
gu = torch.randn(M, 2 * I, device=dev).to(torch.bfloat16)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rep in range(2):
    t_g = timed(lambda: ops.gemm(dy, wd, b_kmajor=True))
    ops.FUSED_SWIGLU_BWD = False
    t_two = timed(lambda: ops.gemm_swiglu_bwd(dy, wd, gu))
    ops.FUSED_SWIGLU_BWD = True
    t_f = timed(lambda: ops.gemm_swiglu_bwd(dy, wd, gu))
    print(f"M={M}: dgrad alone {t_g:6.1f} us | dgrad + swiglu_bwd {t_two:6.1f} us | fused {t_f:6.1f} us", flush=True)
