"""Every distinct bf16 GEMM launch of one training step exactly once (per layer at M = 12336 tokens: four forward, four dgrad
and the grouped weight-gradient launch; + the lm-head trio on 4096 label rows), for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
passes: HBM-side traffic per launch.  o forward runs with its fp32 residual epilogue, down forward too, as in the step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-unigen_amd"))
import torch
from unigen_hip import ops
dev = torch.device("cuda:0")
M = 12336
shapes = {"qkv": (2048, 1536), "o": (1536, 1536), "gu": (17920, 1536), "down": (1536, 8960)}
bufs = []
for name, (N, K) in shapes.items():
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
    gw = torch.zeros(N, K, device=dev)
    bufs.append((x, w, dy, gw))
R, V, H = 4096, 159867, 1536
hn = torch.randn(R, H, device=dev).to(torch.bfloat16)
emb = (torch.randn(V, H, device=dev) * 0.02).to(torch.bfloat16)
logits = torch.empty(R, 159872, dtype=torch.bfloat16, device=dev)
gemb = torch.zeros(V, H, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "all"      # "layers" | "head" | "all": tools/gemm_traffic.py weights the groups 28 : 1
ops._handle()                                            # the stream's workspace exists before the counted launches
torch.cuda.synchronize()
if which in ("layers", "all"):
    cos, sin = ops.rope_tables(771, 128, 1e6, dev)
    bq = torch.zeros(2048, dtype=torch.bfloat16, device=dev)
    gu_saved = torch.randn(M, 17920, device=dev).to(torch.bfloat16)
    for name, (x, w, dy, gw) in zip(shapes, bufs):
        # forward as the step launches it: q/k/v with RoPE in the epilogue, o / down with the fp32 residual epilogue, gate_up with SwiGLU
        if name in ("o", "down"):
            res = torch.randn(M, w.shape[0], device=dev)
            ops.gemm(x, w, out=torch.empty_like(res), epilogue=ops.UG_EPI_RESID, resid=res)
        elif name == "qkv":
            ops.gemm_qkv_rope(x, w, bq, cos, sin, 771, 14, 128)
        else:
            ops.gemm_swiglu(x, w)
        # dgrad: the down projection's with the SwiGLU backward in its epilogue (round 4)
        if name == "down":
            ops.gemm_swiglu_bwd(dy, w, gu_saved)
        else:
            ops.gemm(dy, w, b_kmajor=True)
    ops.gemm_wgrad_group([(dy, x, gw, 1) for x, w, dy, gw in bufs])
if which in ("head", "all"):
    ops.gemm(hn, emb, out=logits, N=V, K=H)
    ops.gemm(logits, hn, out=gemb, M=V, N=H, K=R, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
    ops.gemm(logits, emb, M=R, N=H, K=V, b_kmajor=True)
torch.cuda.synchronize()
