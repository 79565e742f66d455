"""Micro-benchmark of the masked flash attention kernels on the training shape (16 x 771 tokens, 12:2 GQA heads of 128):
forward and backward (dQ + dK/dV + the operand transposes) under the t2i mask of bench.py and a causal mask."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
B, L, H, HKV, hd = int(os.environ.get("B", "16")), int(os.environ.get("L", "771")), 12, 2, 128
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * L, (H + 2 * HKV) * hd, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dout = (torch.randn(B * L, H * hd, device=dev, generator=g) * 0.1).to(torch.bfloat16)
r = torch.arange(L, device=dev)
full = torch.ones(B, L, L, dtype=torch.bool, device=dev)
masks = {"causal": ops.mask_causal(B, L, dev), "full": ops.mask_compress(full)}


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, mb in masks.items():
    frac = 0.5 if name == "causal" else 1.0
    o, lse = ops.attn_fwd(qkv, mb, H, HKV, hd)
    t_f = timed(lambda: ops.attn_fwd(qkv, mb, H, HKV, hd))
    if os.environ.get("ROPE") == "2":       # kernel-trace runs: the fused backward only
        cos, sin = ops.rope_tables(L, hd, 1e6, dev)
        db = torch.zeros((H + 2 * HKV) * hd, device=dev)
        print(f"{name}: fused bwd {timed(lambda: ops.attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd, rope=(cos, sin), dbias=db)):7.1f} us")
        continue
    t_b = timed(lambda: ops.attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd))
    if os.environ.get("ROPE") == "1":       # + RoPE transposed and the bias-gradient sums: fused into the stores, or as the separate passes
        cos, sin = ops.rope_tables(L, hd, 1e6, dev)
        db = torch.zeros((H + 2 * HKV) * hd, device=dev)
        t_fused = timed(lambda: ops.attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd, rope=(cos, sin), dbias=db))

        def sep():
            d = ops.attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd)
            ops.rope_(d, cos, sin, L, H + HKV, hd, backward=True)
            ops.colsum_(d, db)
        print(f"       bwd + rope + bias sums: in the stores {t_fused:7.1f} us, separate passes {timed(sep):7.1f} us (plain bwd {t_b:7.1f})")
    fl = 4.0 * B * H * L * L * hd * frac
    print(f"{name:6s} B={B} L={L}: fwd {t_f:7.1f} us ({fl / t_f / 1e6:6.1f} TF/s)   bwd {t_b:7.1f} us ({2.5 * fl / t_b / 1e6:6.1f} TF/s, 5-matmul count)", flush=True)
