"""Ablations of the 256x256 GEMM main loops (probe builds of the library in gpurun_out/ablate/): the shipped loop forms with
and without the LDS-DMA operand stream (-DUG_GEMM_ABLATE_DMA: only the first three k-tiles are fetched; -DUG_GEMM_ABLATE_SRC:
every DMA is issued but re-fetches one of two cache-resident k-tiles -- the LDS-side cost of the stream without its memory
side; results are wrong by construction), on a long-contraction square shape and the backbone's shapes.  TF/s nominal.
Measured (round 2, two-barrier loop / one-barrier loop):
  shipped         8192^3 1304 / 1324   gate_up fwd 1227 / 1215   down dgrad 1173 / 1206   gate_up wgrad  911 /  973
  cache-resident  8192^3 1492 / 1446   gate_up fwd 1308 / 1269   down dgrad 1248 / 1265   gate_up wgrad 1087 / 1248
  no stream       8192^3 1826 / 1671   gate_up fwd 1566 / 1465   down dgrad 1557 / 1455   gate_up wgrad 1423 / 1431
i.e. the loop itself runs at 94 % of what this part sustains on MFMAs alone (1.94 PF); landing 32 KB of LDS-DMA per k-tile
in the LDS next to 96 KB of fragment reads costs 18 %, the memory side of the stream another 13 %.  Two re-designs of the
loop were measured on the way and dropped: a free-running software-pipelined loop (fragments of tile t+1 re-loaded behind
their last MFMA, one barrier per k-tile, with and without a half-iteration stagger of the two wave groups: 1264 / 1322 on
8192^3, worse on k-major operands and out of registers), and B fragments loaded straight from global memory into registers
(16 bytes per lane, three tiles ahead, 64 KB + 16 KB instead of 96 + 32 KB through the LDS: 688 TF/s on 8192^3 -- per-lane
16-byte loads of 64-byte row pieces run at a fraction of the LDS-DMA rate)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def _probe_build(name, target, flags):
    """tools/probes/build_variant.py: product source + probe_switches.patch, compiled with the given -D flags"""
    bv = os.path.join(ROOT, "tools", "probes", "build_variant.py")
    return subprocess.check_output([sys.executable, bv, name, target, *flags], text=True).strip().splitlines()[-1]

variant = sys.argv[1] if len(sys.argv) > 1 else "full"
flags = {"full": [], "nodma": ["-DUG_GEMM_ABLATE_DMA"], "hotsrc": ["-DUG_GEMM_ABLATE_SRC"],
         "q2": ["-DUG_P10_GROUP_M=2"], "q4": ["-DUG_P10_GROUP_M=4"], "q8": ["-DUG_P10_GROUP_M=8"], "q3": ["-DUG_P10_GROUP_M=3"], "q6": ["-DUG_P10_GROUP_M=6"], "gm2": ["-DUG_P8_GROUP_M=2"], "gm8": ["-DUG_P8_GROUP_M=8"], "gm6": ["-DUG_P8_GROUP_M=6"], "gm16": ["-DUG_P8_GROUP_M=16"]}[variant]
so = _probe_build(f"abl_{variant}", "gemm_bf16.hip", flags)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import lib as L, ops
L.LIB_PATH = so
assert L.load()._name == so
dev = torch.device("cuda:0")
T = 12336
cases = [("sq8192", 8192, 8192, 8192, "fwd"), ("gu_f", T, 17920, 1536, "fwd"), ("down_d", T, 8960, 1536, "dgrad"), ("gu_w", 17920, 1536, T, "wgrad"),
         ("qkv_f", T, 2048, 1536, "fwd"), ("head_f", 4096, 159872, 1536, "fwd"), ("gu_d", T, 1536, 17920, "dgrad"),
         ("qkv_d", T, 1536, 2048, "dgrad"), ("o_f", T, 1536, 1536, "fwd")]
if variant.startswith("q"):
    cases = [c for c in cases if c[0] in ("gu_f", "gu_d", "qkv_d", "o_f")]
pols = {"two-barrier": 105, "one-barrier": 103} if not variant.startswith("q") else {"auto": -1}
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
    best = {k: 0.0 for k in pols}
    for _ in range(3):
        for k, pol in pols.items():
            ops.set_gemm_tile_policy(pol)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record(); torch.cuda.synchronize()
            best[k] = max(best[k], 2.0 * M * N * K * 4 / e0.elapsed_time(e1) / 1e9)
    print(f"[{variant:5s}] {name:7s} " + "  ".join(f"{k}: {v:7.1f}" for k, v in best.items()), flush=True)
ops.set_gemm_tile_policy(-1)
