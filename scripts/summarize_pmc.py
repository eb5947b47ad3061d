#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc counters: python scripts/summarize_pmc.py <dir> [name filter]"""
import csv, glob, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        if filt and filt not in k:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    n = max(len(v) for v in cs.values())
    print(f"{k}  ({n} launches)")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.0f}")
