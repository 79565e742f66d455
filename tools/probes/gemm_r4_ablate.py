"""Ablations of the register-blocked 4-wave GEMM loop (gemm_kernel_r4, policy 12) next to the shipped automatic selection.
Probe builds of the library (tools/probes/_build/, built where hipcc is -- `python tools/probes/gemm_r4_ablate.py build` in the
build container; the .so files travel to the GPU box with the tree):
  full      the kernel as shipped
  nodma     -DUG_R4_ABLATE_DMA      no operand stream after the prologue (results wrong by construction)
  noreads   -DUG_R4_ABLATE_READS    no fragment reads in the loop
  nobar     -DUG_R4_ABLATE_BARRIER  no workgroup barrier in the loop (racy by construction)
  mfmaonly  all three
TF/s nominal, random operands, best of 3 rounds of 4 launches."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "ml-unigen_amd", "csrc")
out = os.path.join(ROOT, "tools", "probes", "_build")
VARIANTS = {"full": [], "nodma": ["-DUG_R4_ABLATE_DMA"], "noreads": ["-DUG_R4_ABLATE_READS"], "nobar": ["-DUG_R4_ABLATE_BARRIER"],
            "mfmaonly": ["-DUG_R4_ABLATE_DMA", "-DUG_R4_ABLATE_READS", "-DUG_R4_ABLATE_BARRIER"]}
VARIANTS = {k: ["-DUG_GEMM_R4"] + v for k, v in VARIANTS.items()}
extra = [a for a in sys.argv[1:] if a.startswith("-D")]


def build(variant):
    bv = os.path.join(ROOT, "tools", "probes", "build_variant.py")          # product source + probe_switches.patch + the flags
    return subprocess.check_output([sys.executable, bv, variant, "gemm_bf16.hip", *VARIANTS[variant], *extra], text=True).strip().splitlines()[-1]


if len(sys.argv) > 1 and sys.argv[1] == "build":
    from concurrent.futures import ThreadPoolExecutor
    names = [a for a in sys.argv[2:] if not a.startswith("-D")] or list(VARIANTS)
    with ThreadPoolExecutor(4) as ex:
        print(list(ex.map(build, names)))
    sys.exit(0)

variant = sys.argv[1] if len(sys.argv) > 1 else "full"
so = os.path.join(out, f"libunigen_hip_{variant}.so")
if not os.path.exists(so):
    build(variant)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import lib as L, ops
L.LIB_PATH = so
assert L.load()._name == so
dev = torch.device("cuda:0")
T = 12336
cases = [("sq8192", 8192, 8192, 8192, "fwd"), ("gu_f", T, 17920, 1536, "fwd"), ("down_d", T, 8960, 1536, "dgrad"), ("gu_w", 17920, 1536, T, "wgrad"),
         ("head_f", 4096, 159872, 1536, "fwd")]
only = os.environ.get("ONLY")
if only:
    cases = [c for c in cases if c[0] in only.split(",")]
pols = {"r4": 12, "auto": -1} if variant == "full" else {"r4": 12}
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
    print(f"[{variant:8s}] {name:7s} " + "  ".join(f"{k}: {v:7.1f}" for k, v in best.items()), flush=True)
ops.set_gemm_tile_policy(-1)
