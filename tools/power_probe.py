"""Package power and shader clock (hwmon: power1_input, freq1_input) sampled every 20 ms while one workload at a time runs for ~4 s:
is the training step bound by the part's power budget, and which kernels are?  (read-only sysfs; ordinary user)

    python3 tools/power_probe.py  > gpurun_out/power_probe.md

Workloads: the bf16 GEMM (gate_up forward, 12 336 x 17 920 x 1 536), the same launch with all-zero operands (no operand toggling),
one with +-0.5 operands, SwiGLU backward and the flat AdamW update (HBM-bound), flash attention forward / backward, the tokenizer
(MAGVITv2.get_code, 16 x 256^2).  The whole training step is sampled by tools/probes/power_sample.sh (bench.py in a child process)."""
import glob
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch
from unigen_hip import ops

hws = [h for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if os.path.exists(os.path.join(h, "power1_input"))]
cap = int(open(os.path.join(hws[0], "power1_cap")).read()) / 1e6
PW = "power1_input of every card; the busiest card of each window is reported"
dev = torch.device("cuda:0")


SAMPLER = """
import sys, time
out, hws = sys.argv[1], sys.argv[2:]
with open(out, "w") as f:
    while True:
        row = []
        for h in hws:
            try:
                row += [open(h + "/power1_input").read().strip(), open(h + "/freq1_input").read().strip()]
            except (OSError, ValueError):
                row += ["0", "0"]
        f.write("%.4f %s\\n" % (time.time(), " ".join(row))); f.flush()
        time.sleep(0.02)
"""
import subprocess
import tempfile
LOG = os.path.join(tempfile.gettempdir(), "power_probe_samples.txt")
sampler = subprocess.Popen([sys.executable, "-c", SAMPLER, LOG] + hws)      # its own process: never waits for this one's GIL


def run(name, fn, seconds=4.0, unit=None, per_call=None):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    one = time.perf_counter() - t0
    reps = max(3, int(seconds / max(one, 1e-5)))
    w0 = time.time()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    w1 = time.time()
    w0 += (w1 - w0) / 4                                         # steady state: drop the ramp
    cards = {}
    for line in open(LOG):
        f = line.split()
        try:
            if len(f) < 3 or len(f) % 2 == 0 or not (w0 <= float(f[0]) <= w1):
                continue
            for c in range((len(f) - 1) // 2):
                cards.setdefault(c, []).append((int(f[1 + 2 * c]) / 1e6, int(f[2 + 2 * c]) / 1e6))
        except ValueError:
            continue
    smp = max(cards.values(), key=lambda v: sum(s[0] for s in v) / len(v))      # the card that ran the workload
    pw = sum(s[0] for s in smp) / len(smp); fq = sum(s[1] for s in smp) / len(smp)
    rate = f"{per_call / dt / 1e12:.0f} {unit}" if per_call else ""
    print(f"| {name} | {dt * 1e3:.3f} | {rate} | {pw:.0f} | {max(s[0] for s in smp):.0f} | {fq:.0f} | {min(s[1] for s in smp):.0f} | {len(smp)} |", flush=True)


print(f"# Power probe ({PW}; cap {cap:.0f} W)\n")
print("| workload | ms per call | rate | mean W | max W | mean sclk MHz | min sclk MHz | samples |\n|---|---|---|---|---|---|---|---|")
M, N, K = 12336, 17920, 1536
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
run("bf16 GEMM gate_up forward, random operands", lambda: ops.gemm(x, w, out=out), unit="TF/s", per_call=2.0 * M * N * K)
xz, wz = torch.zeros_like(x), torch.zeros_like(w)
run("the same launch, all-zero operands", lambda: ops.gemm(xz, wz, out=out), unit="TF/s", per_call=2.0 * M * N * K)
xs = (torch.randn(M, K, device=dev).sign() * 0.5).to(torch.bfloat16)     # one magnitude, random signs: few mantissa / exponent toggles
ws = (torch.randn(N, K, device=dev).sign() * 0.5).to(torch.bfloat16)
run("the same launch, operands +-0.5", lambda: ops.gemm(xs, ws, out=out), unit="TF/s", per_call=2.0 * M * N * K)
del x, w, xz, wz, xs, ws, out

gu = torch.randn(M, 2 * 8960, device=dev).to(torch.bfloat16)
dact = torch.randn(M, 8960, device=dev).to(torch.bfloat16)
run("SwiGLU backward (HBM-bound)", lambda: ops.swiglu_bwd(gu, dact), unit="TB/s", per_call=M * 8960 * 2 * 5.0)
del gu, dact

n = 400_000_000
p, gr, m, v = (torch.zeros(n, device=dev) for _ in range(4))
b16 = torch.zeros(n, dtype=torch.bfloat16, device=dev)
run("flat AdamW, whole chip (HBM-bound)", lambda: ops.adamw_flat_(p, gr, m, v, b16, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3), unit="TB/s", per_call=n * 30.0)
run("flat AdamW, 256 lean workgroups", lambda: ops.adamw_flat_(p, gr, m, v, b16, 1e-4, 0.9, 0.999, 1e-8, 0.01, 3, max_blocks=256), unit="TB/s", per_call=n * 30.0)
del p, gr, m, v, b16

B, L, H, HKV, hd = 16, 771, 12, 2, 128
qkv = (torch.randn(B * L, (H + 2 * HKV) * hd, device=dev) * 0.5).to(torch.bfloat16)
dout = (torch.randn(B * L, H * hd, device=dev) * 0.1).to(torch.bfloat16)
mb = ops.mask_causal(B, L, dev)
o, lse = ops.attn_fwd(qkv, mb, H, HKV, hd)
run("attention forward (causal)", lambda: ops.attn_fwd(qkv, mb, H, HKV, hd))
run("attention backward (causal)", lambda: ops.attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd))
del qkv, dout, o, lse

from models import MAGVITv2
from bench import init_magvit_device
vq = MAGVITv2().to(dev).eval().requires_grad_(False)
init_magvit_device(vq, 10084)
images = torch.rand(16, 3, 256, 256, device=dev) * 2 - 1
run("tokenizer get_code (16 x 256^2)", lambda: vq.get_code(images))

# the matrix cores alone (tools/probes/mfma_peak.hip: v_mfma_f32_16x16x32_bf16, 16 independent accumulators, two waves per SIMD, no operand stream)
import ctypes
so = os.path.join(ROOT, "gpurun_out", "mfma_peak.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "mfma_peak.hip"), "-o", so])
burn = ctypes.CDLL(so)
sink = torch.empty(1 << 20, device=dev)
for data in ("randn", "zeros"):
    seed = (torch.zeros(4096) if data == "zeros" else torch.randn(4096)).to(torch.bfloat16).view(torch.int16).to(dev)
    blocks, threads, iters, nacc = 512, 256, 20000, 16
    args = (ctypes.c_void_p(seed.data_ptr()), ctypes.c_void_p(sink.data_ptr()), blocks, threads, iters, nacc,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    run(f"MFMA only (16x16x32 bf16, two waves per SIMD), {data} operands", lambda: burn.mfma_burn_launch(*args), unit="TF/s",
        per_call=blocks * (threads // 64) * iters * nacc * 2.0 * 16 * 16 * 32)
sampler.terminate()
