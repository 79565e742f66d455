"""Where a decode GEMV's time goes, per wave (probe library from decode_trace_build.py; run with
UNIGEN_HIP_LIB=tools/probes/_build/libunigen_hip_dtrace.so).  One AR generation; the trace buffer keeps the stamps of the LAST
launch of each kernel (last layer of the last step).  Prints, per kernel, the distribution over waves of every stamp relative to
the earliest kernel entry, in us."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from unigen_hip import lib as _l

dev = torch.device("cuda:0")
buf = torch.zeros(3 * 4096 * 8, dtype=torch.int64, device=dev)
L = _l.load()
fn = getattr(L, "ug_decode_trace_set")
fn.argtypes = [ctypes.c_void_p]
assert fn(buf.data_ptr()) == 0

import ar_bench
print(ar_bench.run(use_graph=True, reps=0))
torch.cuda.synchronize()
t = buf.cpu().view(3, 4096, 8)
names = ["gate/up  (RESID_NORM, 9 waves x 3 tiles)", "q/k/v    (RESID_NORM, 4 waves x 1 tile)", "down     (SWIGLU, 8 waves x 2 tiles)"]
stamps = ["entry", "operand + 2 tiles issued", "operand landed", "operand image in LDS (barrier)", "first tile landed",
          "all MFMA done, atomics issued", "atomics acknowledged"]
for r in range(3):
    live = t[r][:, 0] > 0
    raw = t[r][live].double()
    if raw.numel() == 0:
        continue
    # slots 0-6: shader-clock stamps (s_memtime); slot 7: s_memrealtime (100 MHz) at entry.  Shader MHz from the two clocks' spans.
    span_c, span_r = (raw[:, 0].max() - raw[:, 0].min()).item(), (raw[:, 7].max() - raw[:, 7].min()).item()
    mhz = float(os.environ.get("SCLK_MHZ", 0)) or 2100.0
    entry_us = (raw[:, 7] - raw[:, 7].min()) * 0.01
    x = (raw[:, :7] - raw[:, 0:1]) / mhz + entry_us[:, None]
    print(f"(entry spread: {span_r * 0.01:.2f} us by the 100 MHz clock, {span_c:.0f} shader clocks; stamps converted at {mhz:.0f} MHz)")
    print(f"\n## {names[r]}: {x.shape[0]} waves")
    print("| stamp | min | p10 | median | p90 | max |\n|---|---|---|---|---|---|")
    for k in range(7):
        c = x[:, k].sort().values
        q = lambda p: c[min(int(p * (len(c) - 1)), len(c) - 1)].item()
        print(f"| {k} {stamps[k]} | {q(0):.2f} | {q(0.1):.2f} | {q(0.5):.2f} | {q(0.9):.2f} | {q(1.0):.2f} |")
    d = x[:, 1:] - x[:, :-1]
    print("per-wave deltas (median / p90): " + ", ".join(f"{k}->{k+1}: {d[:, k].median().item():.2f} / {d[:, k].sort().values[int(0.9 * (d.shape[0] - 1))].item():.2f}" for k in range(6)))
