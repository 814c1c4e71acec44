"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection.csv.

Usage: python tools/summarize_pmc.py <dir-with-csv> [kernel-substring]
"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if want in k:
            acc[k[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
