"""Micro-benchmark of the fp32 MFMA implicit-GEMM conv on the MAGVITv2 encoder's layer shapes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "16"))
shapes = [(128, 128, 256, 3), (128, 256, 128, 3), (256, 256, 128, 3), (256, 256, 64, 3), (256, 512, 32, 3), (512, 512, 32, 3),
          (512, 512, 16, 3), (128, 256, 128, 1), (4, 128, 256, 3)]
for cin, cout, H, k in shapes:
    x = torch.randn(B, H, H, cin, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    bias = torch.zeros(cout, device=dev)
    wp, cpad = ops.pack_conv_weight(w)
    ops.conv2d_nhwc(x, wp, cpad, bias, cout, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        ops.conv2d_nhwc(x, wp, cpad, bias, cout, k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * H * H * cin * cout * k * k
    line = f"conv {cin:4d}->{cout:4d} k{k} @{H:3d}^2 B={B}: fp32-mfma {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF/s"
    if ops.conv_split_eligible(cin, cout, cpad):
        ws = ops.split_conv_weight(wp)
        ops.conv2d_nhwc(x, wp, cpad, bias, cout, k, w_split=ws)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.conv2d_nhwc(x, wp, cpad, bias, cout, k, w_split=ws)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / reps
        line += f" | f16x2  {ms2:8.3f} ms {fl / ms2 / 1e9:7.1f} TF/s"
        if k == 3:
            ops.conv3x3_nhwc(x, ws, cpad, bias, cout)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                ops.conv3x3_nhwc(x, ws, cpad, bias, cout)
            e1.record()
            torch.cuda.synchronize()
            ms3 = e0.elapsed_time(e1) / reps
            line += f" | patch {ms3:8.3f} ms {fl / ms3 / 1e9:7.1f} TF/s"
    print(line, flush=True)
