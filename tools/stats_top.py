"""Print the top rows of a rocprofv3 *_kernel_stats.csv with kernel names shortened."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:n]:
    name = r["Name"]
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    name = (name[:name.index("(")] if "(" in name else name)[:70]
    print(f"{name:70s} calls={r['Calls']:>8s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={float(r['Percentage']):6.2f}")
