"""AdamW kernel bandwidth: one flat run of 400 M parameters (30 B per parameter with the bf16 mirror)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
n = 400_000_000
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
for _ in range(2):
    ops.adamw_flat_(p, g, m, v, pb, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(5):
    ops.adamw_flat_(p, g, m, v, pb, 1e-4, 0.9, 0.999, 1e-8, 0.01, 2 + i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"adamw {n/1e6:.0f} M params: {ms:.3f} ms  {30.0 * n / ms / 1e9:.2f} TB/s")
