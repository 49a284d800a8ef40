#!/usr/bin/env python3
"""Per-kernel duration summary from a rocprofv3 kernel-trace database (rocpd sqlite):
    python tools/prof_kernels.py results.db [substring]
Prints, per (kernel, grid, LDS bytes), the number of dispatches and the median / min duration."""
import re
import sqlite3
import statistics
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size "
                  f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
acc = defaultdict(list)
order = []
for name, s, e, g, w, lds in rows:
    if sub not in name:
        continue
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
    key = (short, g // max(w, 1), lds)
    if key not in acc:
        order.append(key)
    acc[key].append((e - s) / 1e3)
for key in order:
    v = acc[key]
    print(f"{key[0]:70s} grid {key[1]:7d} lds {key[2]:6d} n={len(v):4d} med {statistics.median(v):9.1f} us  min {min(v):9.1f} us")
