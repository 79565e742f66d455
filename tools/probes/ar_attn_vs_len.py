"""attention-decode kernel time by decode step (context length) from a rocprofv3 kernel trace of tools/ar_bench.py graph"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if "ar_sample" in r[2]]
steps = [rows[a + 1:b + 1] for a, b in zip(idx[:-1], idx[1:])]
steps = [s for s in steps if len(s) == 143][-255:]
for key in sys.argv[2:]:
    print(key)
    for si in range(0, len(steps), 32):
        d = [(e - s) / 1e3 for s, e, n in steps[si] if key in n]
        print(f"  step {si:3d} (context {139 + si}): mean {sum(d) / len(d):6.2f} us  min {min(d):6.2f}  max {max(d):6.2f}")
