#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run: python scripts/summarize_prof.py <dir> [steps] [out.md]"""
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
import os
f = max(glob.glob(d + "/*/*kernel_stats.csv"), key=os.path.getsize)      # (bench.py starts a small child: the python process has the largest trace)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
lines = [f"source: {f}", f"GPU kernel time {tot / 1e6 / steps:.2f} ms/step, {calls / steps:.0f} launches/step (over {steps:g} steps incl. warm-up)", "",
         "| ms/step | % | calls/step | avg us | kernel |", "|---|---|---|---|---|"]
for r in rows[:40]:
    lines.append(f"| {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['Percentage']):.2f} | {int(r['Calls']) / steps:.1f} | "
                 f"{float(r['AverageNs']) / 1e3:.1f} | `{r['Name'][:110]}` |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
