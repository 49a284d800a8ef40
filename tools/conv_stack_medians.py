#!/usr/bin/env python3
"""Per-layer, per-kernel median durations of a rocprofv3 --kernel-trace run of tools/conv_stack_trace.py: a layer starts
where a weight-packing kernel follows a convolution kernel.    python tools/conv_stack_medians.py DIR"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
layers, cur, seen_conv = [], {}, False
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if name.startswith("at::"):
        continue
    is_pack = "pack" in name and "prep" not in name or (("patch_w8_prep" in name or "stem_w8_prep" in name) and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) < 400000 and seen_conv)
    if is_pack and seen_conv:
        layers.append(cur)
        cur, seen_conv = {}, False
    if not ("pack" in name and "prep" not in name):
        cur.setdefault(name[:80], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if "prep" not in name and "pad" not in name:
            seen_conv = True
layers.append(cur)
tot = 0.0
for i, lay in enumerate(layers):
    line, s = [], 0.0
    for k, v in lay.items():
        v.sort()
        med = v[len(v) // 2] / 1e3
        if len(v) >= 4:
            s += med
            line.append(f"{k} {med:.1f}")
    print(f"layer {i}: sum {s:6.1f} us | " + " | ".join(line))
