"""HBM-side bytes per bf16 GEMM launch of one training step, from rocprofv3 PMC passes over tools/gemm_step_mix.py
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes, so it is doubled; the counters are in KiB).  Four passes on the GPU box:

    for g in layers head; do for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${g}_$c -- python3 tools/gemm_step_mix.py $g; done; done
    python3 tools/gemm_traffic.py gpurun_out            # writes gpurun_out/gemm_traffic_current.json

The record carries the sha256 of the gemm_bf16.hip it was measured on; bench.py reports `roofline.traffic` only while that
still matches the source it runs (a stale record reads as null).  Copy the file to profiles/ to commit it."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
WEIGHT = {"layers": 28, "head": 1}


def kernel_rows(group, counter):
    files = glob.glob(os.path.join(base, f"pmc_{group}_{counter}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {base}/pmc_{group}_{counter}")
    rows = []
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter and ("gemm_kernel" in r["Kernel_Name"] or "tail_finish_kernel" in r["Kernel_Name"]):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(rows)


per, tot_bytes, launches = [], 0.0, 0
for group, w in WEIGHT.items():
    fetch, write = kernel_rows(group, "FETCH_SIZE"), kernel_rows(group, "WRITE_SIZE")
    assert len(fetch) == len(write), (group, len(fetch), len(write))
    for (_, name, f), (_, name2, wr) in zip(fetch, write):
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        b = (2.0 * f + wr) * 1024.0
        per.append({"group": group, "kernel": short, "fetch_KiB_raw": f, "write_KiB": wr, "bytes": b})
        tot_bytes += w * b
        if "tail_finish" not in short:
            launches += w                                   # a finishing pass belongs to its GEMM launch
src = open(os.path.join(ROOT, "ml-unigen_amd", "csrc", "gemm_bf16.hip"), "rb").read()
rec = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/gemm_step_mix.py {layers,head}; "
                 "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; step = 28 x the 9 per-layer launches (4 forward, 4 dgrad, the grouped weight gradients) + the lm-head trio",
       "gemm_src_sha256": hashlib.sha256(src).hexdigest(), "launches_per_step": launches,
       "traffic_bytes_per_step": tot_bytes, "traffic_bytes_per_launch": round(tot_bytes / launches), "per_dispatch": per}
out = os.path.join(base, "gemm_traffic_current.json")
json.dump(rec, open(out, "w"), indent=1)
print(f"{launches} launches/step, {tot_bytes / 1e9:.1f} GB/step, {tot_bytes / launches / 1e6:.1f} MB per launch -> {out}")
