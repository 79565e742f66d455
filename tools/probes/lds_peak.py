"""Run tools/probes/lds_peak.hip: LDS fragment-read bandwidth per CU (bytes/clock at the reported shader clock), alone and
with 1-4 bf16 MFMAs consuming every fragment -- the ceiling a GEMM's wave tiling has to live under."""
import ctypes
import os
import subprocess
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(root, "gpurun_out", "lds_peak.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                       os.path.join(root, "tools", "probes", "lds_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
out = torch.empty(1 << 20, device=dev)
mhz = 2400.0     # MI355X peak engine clock; sustained clocks under load are lower
print(f"bytes/clock quoted at {mhz:.0f} MHz")
for threads in (256, 512):
    for mpr in (0, 1, 2, 3, 4, 100, 102, 103):
        iters = 4000
        args = (ctypes.c_void_p(out.data_ptr()), 256, threads, iters, mpr, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        lib.lds_read_launch(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.lds_read_launch(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        waves = threads // 64
        byts = 256 * waves * iters * 16 * 1024.0
        flops = 256 * waves * iters * 16 * (mpr % 100) * 2.0 * 16 * 16 * 32
        print(f"waves/CU={waves} {'tr16_b64 x2' if mpr >= 100 else 'b128'} mfma/read={mpr % 100}: LDS {byts / ms / 1e9:7.1f} TB/s = {byts / 256 / (ms * 1e-3) / (mhz * 1e6):6.1f} B/clk/CU"
              f"   MFMA {flops / ms / 1e9:7.1f} TF/s ({ms:.2f} ms)", flush=True)
