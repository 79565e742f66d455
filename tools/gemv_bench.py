"""Decode GEMV micro-benchmark: the four projections of one Qwen2.5-1.5B layer at 16 rows, rotating over 28
independent weight buffers so nothing is served from L2 / Infinity Cache.  Prints us per launch and GB/s."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
R = int(os.environ.get("ROWS", 16))
SHAPES = [("qkv", 2048, 1536), ("o", 1536, 1536), ("gu", 17920, 1536), ("down", 1536, 8960), ("head", 8192, 1536)]
for name, N, K in SHAPES:
    ws = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(28)]
    x = torch.randn(R, K, device=dev).to(torch.bfloat16)
    acc = torch.zeros(R, N, device=dev)
    for w in ws[:3]:
        ops.gemv_acc_(x, w, acc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    # replay a captured graph so the host launch rate (~7.5 us per ctypes call) does not floor the small shapes
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph):
            for w in ws:
                ops.gemv_acc_(x, w, acc)
    graph.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * len(ws))
    print(f"{name:5s} N={N:6d} K={K:5d}  {us:7.2f} us  {N * K * 2 / us / 1e3:8.1f} GB/s", flush=True)
