"""Independent yardstick for the bf16 GEMM: the same backbone shapes through the library's kernel and through torch.matmul
(hipBLASLt / rocBLAS as shipped with PyTorch-ROCm), alternating, HIP events, random operands.  Not on the product path."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "12336"))
shapes = {"qkv": (2048, 1536), "o": (1536, 1536), "gate_up": (17920, 1536), "down": (1536, 8960)}
reps = int(os.environ.get("REPS", "10"))


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"M = {M}; TF/s (ours | torch.matmul); random bf16 operands, {reps} launches each")
tot_o = tot_t = 0.0
for name, (N, K) in shapes.items():
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    gw = torch.zeros(N, K, device=dev)
    gwt = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    out_f = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out_d = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    line = []
    for mode in ("fwd", "dgrad", "wgrad"):
        if mode == "fwd":
            ours = lambda: ops.gemm(x, w)
            ref = lambda: torch.matmul(x, w.t(), out=out_f)
        elif mode == "dgrad":
            ours = lambda: ops.gemm(dy, w, b_kmajor=True)
            ref = lambda: torch.matmul(dy, w, out=out_d)
        else:
            ours = lambda: ops.gemm(dy, x, out=gw, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
            ref = lambda: torch.matmul(dy.t(), x, out=gwt)            # bf16 output, no accumulate: LESS work than ours
        to, tt = timed(ours), timed(ref)
        to2, tt2 = timed(ours), timed(ref)
        to, tt = min(to, to2), min(tt, tt2)
        tot_o += to
        tot_t += tt
        fl = 2.0 * M * N * K / 1e9
        line.append(f"{mode} {fl / to:7.1f} | {fl / tt:7.1f}")
    print(f"{name:8s} N={N:6d} K={K:5d}: " + "   ".join(line), flush=True)
print(f"sum of the twelve launches: ours {tot_o:.3f} ms, torch.matmul {tot_t:.3f} ms")
# a big square for reference
for n in (8192,):
    a = torch.randn(n, n, device=dev).to(torch.bfloat16)
    b = torch.randn(n, n, device=dev).to(torch.bfloat16)
    o = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
    to, tt = timed(lambda: ops.gemm(a, b)), timed(lambda: torch.matmul(a, b.t(), out=o))
    print(f"{n}^3: ours {2.0 * n ** 3 / to / 1e9:7.1f} | torch.matmul {2.0 * n ** 3 / tt / 1e9:7.1f} TF/s")
