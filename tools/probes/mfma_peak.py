"""Run tools/probes/mfma_peak.hip: chip-wide bf16 MFMA rate with zero and with random operands, 1 and 2 waves/SIMD."""
import ctypes
import os
import subprocess
import sys
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(root, "gpurun_out", "mfma_peak.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                       os.path.join(root, "tools", "probes", "mfma_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
out = torch.empty(1 << 20, device=dev)
for data in ("zeros", "randn"):
    seed = (torch.zeros(4096) if data == "zeros" else torch.randn(4096)).to(torch.bfloat16).view(torch.int16).to(dev)
    for threads, blocks_per_cu in ((256, 1), (256, 2)):
        for nacc in (64, 16):
            blocks, iters = 256 * blocks_per_cu, 20000 if nacc == 16 else 5000
            args = (ctypes.c_void_p(seed.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, threads, iters, nacc,
                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            lib.mfma_burn_launch(*args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.mfma_burn_launch(*args)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            flops = blocks * (threads // 64) * iters * nacc * 2.0 * 16 * 16 * 32
            print(f"{data:6s} threads={threads} blocks/CU={blocks_per_cu} nacc={nacc}: {flops / ms / 1e9:8.1f} TF/s ({ms:.2f} ms)", flush=True)


# v_mfma_f32_32x32x16_{bf16,f16}: 1 and 2 waves per SIMD, 4 and 16 independent accumulator blocks
for data in ("zeros", "randn"):
    seed = (torch.zeros(4096) if data == "zeros" else torch.randn(4096)).to(torch.bfloat16).view(torch.int16).to(dev)
    for f16 in (0, 1):
        if f16:
            seed = (torch.zeros(4096) if data == "zeros" else torch.randn(4096)).to(torch.float16).view(torch.int16).to(dev)
        for blocks_per_cu in (1, 2):
            for nacc in (16, 4):
                if nacc == 16 and blocks_per_cu == 2 and False:
                    continue
                blocks, iters = 256 * blocks_per_cu, 4000 if nacc == 16 else 16000
                args = (ctypes.c_void_p(seed.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, 256, iters, nacc, f16,
                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                lib.mfma_burn32_launch(*args)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.mfma_burn32_launch(*args)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
                flops = blocks * 4 * iters * nacc * 2.0 * 32 * 32 * 16
                cyc = ms * 1e-3 * 2.4e9 / (iters * nacc * blocks_per_cu)
                print(f"32x32x16 {'f16 ' if f16 else 'bf16'} {data:6s} waves/SIMD={blocks_per_cu} nacc={nacc}: {flops / ms / 1e9:8.1f} TF/s "
                      f"({ms:.2f} ms; {cyc:.1f} clocks per MFMA per SIMD at 2.4 GHz)", flush=True)

# fp32-input MFMA (32x32x2): 1, 2 and 4 waves per SIMD
seed = torch.randn(4096).mul(1000).to(torch.int16).to(dev)
for blocks_per_cu in (1, 2, 4):
    blocks, iters = 256 * blocks_per_cu, 20000
    args = (ctypes.c_void_p(seed.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, 256, iters,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    lib.mfma_burn_f32_launch(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.mfma_burn_f32_launch(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flops = blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2
    print(f"fp32 32x32x2 blocks/CU={blocks_per_cu}: {flops / ms / 1e9:8.1f} TF/s ({ms:.2f} ms)", flush=True)
