"""Where the optimizer update runs relative to the next step's tokenisation and forward, from a rocprofv3 kernel trace of bench.py:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ov -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 4
    python3 tools/overlap_timeline.py gpurun_out/prof_ov

For each optimizer update (a run of adamw kernels) of the timed steps: its start and end relative to the first tokenizer kernel that
follows the backward, the tokenizer's own span, how much of the update ran inside that span, and which kernels of the main stream ran
while the update's tail was still in flight."""
import csv
import glob
import os
import sys

base = sys.argv[1]
files = glob.glob(os.path.join(base, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    raise SystemExit(f"no kernel_trace.csv under {base}")
rows = []
for r in csv.DictReader(open(files[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
is_adam = lambda n: "adamw" in n
is_tok = lambda n: any(k in n for k in ("conv", "groupnorm", "gn_", "lfq", "amax", "nchw", "nhwc"))


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]


# runs of adamw kernels (allowing other kernels interleaved in time): group by gaps > 20 ms between adamw launches
adam = [r for r in rows if is_adam(r[2])]
runs, cur = [], []
for r in adam:
    if cur and r[0] - cur[-1][1] > 20e6:
        runs.append(cur); cur = []
    cur.append(r)
if cur:
    runs.append(cur)
print(f"# {len(runs)} optimizer updates in {files[0].split('/')[-1]}\n")
print("| update | kernels | span ms | busy ms | tokenizer span ms | update inside tokenizer span ms | update tail after tokenizer ms | main-stream kernels under the tail |")
print("|---|---|---|---|---|---|---|---|")
for i, run in enumerate(runs[:-1] if len(runs) > 1 else runs):
    a0, a1 = run[0][0], max(r[1] for r in run)
    busy = sum(r[1] - r[0] for r in run) / 1e6
    # tokenizer kernels that start after the update starts and before the update ends + 40 ms
    tok = [r for r in rows if is_tok(r[2]) and r[0] >= a0 - 1e6 and r[0] <= a0 + 60e6]
    if not tok:
        continue
    # tokenizer span: contiguous cluster starting at the first such kernel (stop at a gap > 3 ms = the next step)
    span = [tok[0]]
    for r in tok[1:]:
        if r[0] - span[-1][1] > 3e6:
            break
        span.append(r)
    t0, t1 = span[0][0], max(r[1] for r in span)
    inside = sum(max(0, min(r[1], t1) - max(r[0], t0)) for r in run) / 1e6
    tail = max(0, a1 - t1) / 1e6
    under = {}
    for r in rows:
        if not is_adam(r[2]) and r[0] < a1 and r[1] > t1:
            k = short(r[2]); under[k] = under.get(k, 0) + 1
    top = ", ".join(f"{k} x{v}" for k, v in sorted(under.items(), key=lambda kv: -kv[1])[:4])
    print(f"| {i} | {len(run)} | {(a1 - a0) / 1e6:.2f} | {busy:.2f} | {(t1 - t0) / 1e6:.2f} (starts {(t0 - a0) / 1e6:+.2f} after the update's first kernel) | {inside:.2f} | {tail:.2f} | {top} |")
