import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
n = 400_000_000
p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
b16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mb in (0, 128, 256, 512, 1024, 2048):
    t = timed(lambda: ops.adamw_flat_(p, g, m, v, b16, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, max_blocks=mb))
    print(f"max_blocks={mb}: {t:.3f} ms = {n*30/t/1e9:.2f} TB/s", flush=True)
