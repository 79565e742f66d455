"""Phase timeline of the 16-row patch convolution (workgroup 0, waves 0 and 4): probe build with -DUG_CONV_TRACE in
gpurun_out/ctrace/, one 128 -> 128 convolution at 256^2, median clocks (s_memtime) per tap of each group's phases.
Measured (round 2): L 476 | barrier 602 | M 920 | barrier 156 = 2148 clocks per tap.  A one-barrier-per-tap pipeline (group 0 runs
L(t) M(t), group 1 M(t-1) L(t) with its fragments kept across the barrier and the higher MFMA priority, carried through the slab
boundaries) showed 1956 clocks per tap under the stamps but 0.887 ms against 0.855 ms for the whole 128 -> 128 @256^2 launch
without them, and was not kept."""
import ctypes
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def _probe_build(name, target, flags):
    """tools/probes/build_variant.py: product source + probe_switches.patch, compiled with the given -D flags"""
    bv = os.path.join(ROOT, "tools", "probes", "build_variant.py")
    return subprocess.check_output([sys.executable, bv, name, target, *flags], text=True).strip().splitlines()[-1]

so = _probe_build("ctrace", "conv_split.hip", ["-DUG_CONV_TRACE"])
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import numpy as np
import torch
from unigen_hip import lib as L, ops
L.LIB_PATH = so
assert L.load()._name == so
dev = torch.device("cuda:0")
x = torch.randn(16, 256, 256, 128, device=dev)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.05
wp, cpad = ops.pack_conv_weight(w)
ws = ops.split_conv_weight(wp)
bias = torch.zeros(128, device=dev)
for _ in range(2):
    ops.conv3x3_nhwc(x, ws, cpad, bias, 128)
torch.cuda.synchronize()
buf = np.zeros(2 * 64 * 4, dtype=np.uint64)
assert ctypes.CDLL(so).ug_conv_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
tr = buf.reshape(2, 64, 4).astype(np.int64)
for grp in (0, 1):
    t = tr[grp, 1:35]                       # taps 1..34 of the 36 (4 slabs x 9)
    Lp, b1, Mp = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    nxt = t[1:, 0] - t[:-1, 3]
    per = t[1:, 0] - t[:-1, 0]
    med = lambda a: float(np.median(a))
    print(f"group {grp}: L {med(Lp):5.0f}  barrier {med(b1):5.0f}  M {med(Mp):5.0f}  barrier(+slab work) {med(nxt):5.0f}   per tap {med(per):6.0f} "
          f"(p10 {np.percentile(per, 10):.0f}, p90 {np.percentile(per, 90):.0f}) clocks; 96 MFMAs/SIMD = 1536 at 16 clk", flush=True)
