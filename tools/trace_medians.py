#!/usr/bin/env python3
"""Median duration per kernel (and grid size) of a rocprofv3 --kernel-trace run, in order of first launch:
    python tools/trace_medians.py DIR [substring ...]"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seen = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    if sys.argv[2:] and not any(s in name for s in sys.argv[2:]):
        continue
    seen.setdefault((name[:90], r.get("Grid_Size_X", r.get("Grid_Size", "?"))), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, grid), v in seen.items():
    v.sort()
    print(f"{name:90s} grid {grid:>9s} n={len(v):4d} median {v[len(v) // 2] / 1e3:8.1f} us")
