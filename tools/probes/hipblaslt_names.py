"""Which hipBLASLt kernels torch.matmul picks on the backbone's shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
dev = torch.device("cuda:0")
M = 12336
for (m, n, k, tag) in ((M, 17920, 1536, "gate_up fwd NT"), (8192, 8192, 8192, "square NT"), (M, 1536, 8960, "down fwd NT")):
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    o = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, b.t(), out=o)
    torch.cuda.synchronize()
