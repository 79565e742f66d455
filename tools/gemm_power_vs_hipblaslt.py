"""the library's bf16 GEMM and hipBLASLt (through torch.matmul) on the same shapes UNDER THE POWER PROBE --
TF/s, package watts, shader MHz and pJ per flop for each, ~3 s per measurement with random operands (the step is power-bound:
at the same 1400 W the faster kernel is the one that moves fewer joules per flop).  bench.py's PowerSampler reads hwmon."""
import importlib.util
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(bench)
sys.argv = argv
dev = torch.device("cuda:0")


def measure(name, fn, flops, seconds=3.0):
    sampler = bench.PowerSampler()               # (one sampler per measurement: report() ends its child process)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    reps = max(5, int(seconds / max(time.perf_counter() - t0, 1e-5)))
    w0, t0 = time.time(), time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    pw = sampler.report(w0, time.time()) or {}
    tf = flops / dt / 1e12
    w = pw.get("mean_w")
    print(f"| {name} | {tf:.0f} | {w if w is None else round(w)} | {pw.get('mean_sclk_mhz')} | {None if not w else round(w / (tf * 1e12) * 1e12, 3)} |", flush=True)
    return tf


print("| launch | TF/s | package W | sclk MHz | pJ / flop (package) |\n|---|---|---|---|---|")
for tag, M, N, K in (("8192^3", 8192, 8192, 8192), ("qkv fwd 12336 x 2048 x 1536", 12336, 2048, 1536), ("gate_up fwd 12336 x 17920 x 1536", 12336, 17920, 1536),
                     ("down fwd 12336 x 1536 x 8960", 12336, 1536, 8960)):
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    measure(f"{tag}: library", lambda: ops.gemm(a, b, out=out), fl)
    measure(f"{tag}: hipBLASLt", lambda: torch.matmul(a, b.t(), out=out), fl)
    measure(f"{tag}: library (again)", lambda: ops.gemm(a, b, out=out), fl)
