"""Idle time between consecutive kernels of a rocprofv3 kernel trace: python profiles/exp/trace_gaps.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
gaps = collections.defaultdict(list)
prev = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]
    if prev is not None:
        g = int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])
        if g < 100000:
            gaps[(prev["Kernel_Name"].split("(")[0].replace("void ", "")[:30], n)].append(g)
    prev = r
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 8:
        print(f"{k[0]:32s} -> {k[1]:32s} n={len(v):4d} mean gap {sum(v) / len(v) / 1000:6.2f} us  min {min(v) / 1000:6.2f}")
