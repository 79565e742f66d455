"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (substring filter)."""
import collections
import csv
import sys
path, flt = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if flt and not any(f in name for f in flt):
        continue
    key = name.replace("void ", "").replace("(anonymous namespace)::", "")
    key = (key[:key.index("(")] if "(" in key else key)[:60]
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(key, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:32s} {x:14.4e}  ({n[(k, c)]} dispatches)")
