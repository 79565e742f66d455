"""The three tied-head GEMMs of a training step (4096 label rows x 159 867 vocabulary x 1536)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
R, V, H = 4096, 159867, 1536
hn = torch.randn(R, H, device=dev).to(torch.bfloat16)
emb = (torch.randn(V, H, device=dev) * 0.02).to(torch.bfloat16)
logits = torch.randn(R, 159872, device=dev).to(torch.bfloat16)
gemb = torch.zeros(V, H, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * R * V * H
for name, fn in (("fwd", lambda: ops.gemm(hn, emb, out=logits, N=V, K=H)),
                 ("wgrad", lambda: ops.gemm(logits, hn, out=gemb, M=V, N=H, K=R, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)),
                 ("dgrad", lambda: ops.gemm(logits, emb, M=R, N=H, K=V, b_kmajor=True))):
    ms = t(fn)
    print(f"head {name}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s", flush=True)
