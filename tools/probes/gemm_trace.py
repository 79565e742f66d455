"""Phase timeline of the staggered 256x256 GEMM (workgroup 0, waves 0 and 4): builds the library with -DUG_GEMM_TRACE into
gpurun_out/, runs one launch per shape and prints, in shader clocks (s_memtime), the median duration of
  L   = fragment reads + DMA issue + counted vmcnt / lgkmcnt waits        (stamp 0 -> 1)
  b1  = wait at the barrier that opens the M phase                        (1 -> 2)
  M   = issue of the 32 MFMAs                                             (2 -> 3)
  b2  = wait at the barrier that closes it                                (3 -> next 0)
for both wave groups."""
import ctypes
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def _probe_build(name, target, flags):
    """tools/probes/build_variant.py: product source + probe_switches.patch, compiled with the given -D flags"""
    bv = os.path.join(ROOT, "tools", "probes", "build_variant.py")
    return subprocess.check_output([sys.executable, bv, name, target, *flags], text=True).strip().splitlines()[-1]

so = _probe_build("gtrace", "gemm_bf16.hip", ["-DUG_GEMM_TRACE"])
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import numpy as np
import torch
from unigen_hip import lib as L, ops
L.LIB_PATH = so                                   # the probe build instead of the shipped library
assert L.load()._name == so, L.load()._name
dev = torch.device("cuda:0")
T = 12336
cases = [("gu_f", T, 17920, 1536, "fwd"), ("sq8192", 8192, 8192, 8192, "fwd"), ("down_d", T, 8960, 1536, "dgrad"), ("gu_w", 17920, 1536, T, "wgrad")]
raw = ctypes.CDLL(so)
for name, M, N, K, mode in cases:
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    if mode == "fwd":
        a, b = rnd(M, K), rnd(N, K); run = lambda: ops.gemm(a, b)
    elif mode == "dgrad":
        a, b = rnd(M, K), rnd(K, N); run = lambda: ops.gemm(a, b, b_kmajor=True)
    else:
        a, b = rnd(K, M), rnd(K, N); o32 = torch.zeros(M, N, device=dev)
        run = lambda: ops.gemm(a, b, out=o32, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
    for pol, label in ((3, "two barriers"), (103, "one barrier ")):
        ops.set_gemm_tile_policy(pol)
        run(); torch.cuda.synchronize(); run(); torch.cuda.synchronize()
        buf = np.zeros(2 * 512 * 4, dtype=np.uint64)
        assert raw.ug_gemm_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        tr = buf.reshape(2, 512, 4).astype(np.int64)
        nk = min(512, (K + 31) // 32)
        for grp in (0, 1):
            t = tr[grp, 4:nk - 4]
            d01, d12, d23 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
            d30 = t[1:, 0] - t[:-1, 3]
            per = t[1:, 0] - t[:-1, 0]
            med = lambda x: float(np.median(x))
            if pol == 3:
                txt = f"L {med(d01):5.0f}  barrier {med(d12):5.0f}  M {med(d23):5.0f}  barrier {med(d30):5.0f}"
            elif grp == 0:
                txt = f"L {med(d01):5.0f}  M {med(d12):5.0f}  dma-wait {med(d23):5.0f}  barrier {med(d30):5.0f}"
            else:
                txt = f"M {med(d01):5.0f}  L {med(d12):5.0f}  dma-wait {med(d23):5.0f}  barrier {med(d30):5.0f}"
            print(f"{name:7s} {label} group {grp}: {txt}   iteration {med(per):6.0f} (p10 {np.percentile(per, 10):.0f}, p90 {np.percentile(per, 90):.0f})",
                  flush=True)
ops.set_gemm_tile_policy(-1)
