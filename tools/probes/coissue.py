"""Run tools/probes/coissue.hip: do MFMAs and VALU instructions of two waves on one SIMD overlap?"""
import ctypes
import os
import subprocess
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(root, "gpurun_out", "coissue.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                       os.path.join(root, "tools", "probes", "coissue.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
out = torch.empty(1 << 20, device=dev)
seed = torch.randn(4096).to(torch.bfloat16).view(torch.int16).to(dev)
iters = 20000
names = {0: "MFMA on both waves of a SIMD", 1: "VALU on both waves", 2: "MFMA wave + VALU wave per SIMD", 3: "MFMA wave alone (second wave idle)",
         4: "VALU wave alone (first wave idle)"}
for valu_exp in (0, 1):
    print("VALU stream:", "v_fma + v_exp (quarter rate)" if valu_exp else "v_fma + v_fma")
    for mode in (3, 4, 2, 0, 1):
        args = (ctypes.c_void_p(seed.data_ptr()), ctypes.c_void_p(out.data_ptr()), 256, iters, mode, valu_exp,
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        lib.coissue_launch(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.coissue_launch(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"  {names[mode]:42s} {ms:7.2f} ms = {ms * 1e6 / iters:7.1f} ns per iteration (4 MFMAs | 32 VALU instructions)", flush=True)
