"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch (sum over dimensions)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> per-dispatch values
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            key = (k, row.get("Dispatch_Id"), row.get("Counter_Name"))
            per[key] += float(row.get("Counter_Value", 0) or 0)
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
for k in sorted(acc):
    print(f"--- {k}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"  {c:28s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
