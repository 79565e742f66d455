"""rocprofv3 *_kernel_stats.csv -> markdown table (top N kernels), for profiles/*.md."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("| kernel | calls | total ms | avg us | % |")
print("|---|---|---|---|---|")
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"]) / 1e6
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    name = (name[:name.index("(")] if "(" in name else name)[:80]
    print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
print(f"\nSum of kernel time: {tot:.1f} ms")
