"""HBM-bound elementwise kernels of the backbone at the training shape (12 336 tokens): achieved TB/s on their algorithmic
bytes.  Measured (round 2): rmsnorm_fwd 27.9 us = 4.1 TB/s, rmsnorm_bwd (+ bf16 copy of the residual gradient) 62.7 us =
4.8 TB/s (55.5 us without its 1.2 M dw atomics), swiglu_fwd 128.5 us = 5.2 TB/s, swiglu_bwd 215.3 us = 5.1 TB/s."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
T, H, I = 12336, 1536, 8960


COLD = os.environ.get("COLD") == "1"        # evict L2 / Infinity Cache between calls (what a kernel inside the step sees)
_junk = torch.empty(1 << 28, device=dev) if COLD else None          # 1 GiB


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    if COLD:
        tot = 0.0
        for _ in range(reps):
            _junk.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(T, H, device=dev)
w = torch.randn(H, device=dev)
y, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
us = timed(lambda: ops.rmsnorm_fwd(x, w, 1e-6))
print(f"rmsnorm_fwd              {us:7.1f} us  {T * H * 6 / us / 1e6:5.2f} TB/s", flush=True)
dy = torch.randn(T, H, device=dev).to(torch.bfloat16)
dres = torch.randn(T, H, device=dev)
dw = torch.zeros(H, device=dev)
us = timed(lambda: ops.rmsnorm_bwd(dy, x, rstd, w, dres, dw, want_bf16=True))
byts = T * H * (4 + 2 + 4 + 4 + 2)
print(f"rmsnorm_bwd (+bf16 copy) {us:7.1f} us  {byts / us / 1e6:5.2f} TB/s on {byts / 1e6:.0f} MB", flush=True)
gu = torch.randn(T, 2 * I, device=dev).to(torch.bfloat16)
us = timed(lambda: ops.swiglu_fwd(gu))
print(f"swiglu_fwd               {us:7.1f} us  {T * I * 6 / us / 1e6:5.2f} TB/s", flush=True)
dact = torch.randn(T, I, device=dev).to(torch.bfloat16)
us = timed(lambda: ops.swiglu_bwd(gu, dact))
print(f"swiglu_bwd               {us:7.1f} us  {T * I * 10 / us / 1e6:5.2f} TB/s", flush=True)
