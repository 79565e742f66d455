"""Package power against read bandwidth for a streaming kernel whose working set lives in the L2 (per XCD), in the Infinity Cache, or in
HBM (tools/probes/mem_energy.hip): what the GEMMs' L2-miss re-reads of their operand panels cost in joules."""
import ctypes
import glob
import os
import subprocess
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-unigen_amd"))
import torch

so = os.path.join(ROOT, "gpurun_out", "mem_energy.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "mem_energy.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
hws = [h for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if os.path.exists(os.path.join(h, "power1_input"))]
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
LOG = os.path.join(tempfile.gettempdir(), "mem_energy_samples.txt")
sampler = subprocess.Popen([sys.executable, "-c", SAMPLER, LOG] + hws)
buf = torch.randint(0, 2 ** 31 - 1, (8 * 1024 * 1024 * 1024 // 4,), dtype=torch.int32, device=dev)      # 8 GiB
sink = torch.zeros(1 << 16, dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(name, total, span, passes, xcd_local, blocks=2048, seconds=4.0):
    call = lambda: lib.stream_read_launch(ctypes.c_void_p(buf.data_ptr()), ctypes.c_uint64(total), ctypes.c_uint64(span), passes, xcd_local,
                                          blocks, ctypes.c_void_p(sink.data_ptr()), st)
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter(); call(); torch.cuda.synchronize()
    one = time.perf_counter() - t0
    reps = max(3, int(seconds / one))
    w0 = time.time(); t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    w1 = time.time(); w0 += (w1 - w0) / 4
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
    smp = max(cards.values(), key=lambda v: sum(s[0] for s in v) / len(v))
    pw = sum(s[0] for s in smp) / len(smp); fq = sum(s[1] for s in smp) / len(smp)
    bw = blocks * span * passes / dt
    print(f"| {name} | {bw / 1e12:.2f} | {pw:.0f} | {fq:.0f} | {(pw - IDLE) / bw * 1e12:.1f} |", flush=True)


time.sleep(2.0)
idle = []
for line in open(LOG):
    f = line.split()
    if len(f) >= 3 and len(f) % 2 == 1:
        idle.append(min(int(f[1 + 2 * c]) / 1e6 for c in range((len(f) - 1) // 2) if int(f[1 + 2 * c]) > 0))
IDLE = sorted(idle)[len(idle) // 2] if idle else 240.0
print(f"# Memory-level energy probe (idle package power {IDLE:.0f} W; pJ per byte = (package W - idle W) / bytes per second)\n")
print("| working set | read TB/s | package W | sclk MHz | pJ per byte above idle |\n|---|---|---|---|---|")
MB = 1 << 20
run("L2-resident: every XCD re-reads its own 2 MB (16 MB chip-wide)", 16 * MB, 2 * MB, 64, 1)
run("L2-resident, 1 MB per XCD", 8 * MB, 1 * MB, 128, 1)
run("Infinity-Cache-resident: 128 MB walked by consecutive workgroups, 64 KB each per pass", 128 * MB, 64 * 1024, 64, 0)
run("Infinity-Cache-resident: 64 MB", 64 * MB, 32 * 1024, 128, 0)
run("HBM: 8 GiB walked once per launch", 8 * 1024 * MB, 4 * MB, 1, 0)
sampler.terminate()
