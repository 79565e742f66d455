"""print calls / avg us of kernels whose name contains any of the given substrings, from a rocprofv3 kernel_stats.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s}  avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
