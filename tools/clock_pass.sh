#!/bin/bash
# One rocprofv3 counter pass (clock and matrix-pipe occupancy) over a program; prints per dispatch group.
# Usage: tools/clock_pass.sh OUTDIR KERNEL_SUBSTRING GROUP_SIZE -- python3 prog args...
set -u
out=$1; kern=$2; shift 4
export TMPDIR=/tmp
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out -o p -- "$@" > $out.log 2>&1 || { echo "pass failed"; tail -5 $out.log; exit 1; }
grep -v "^\[" $out.log | grep "us " 
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = collections.defaultdict(dict)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows, key=int)
# print a line every time the clock regime changes a lot, plus periodic samples
acc = []
for n, i in enumerate(ids):
    c = rows[i]; d = dur.get(i)
    if not d: continue
    ghz = c["GRBM_GUI_ACTIVE"] / 8 / d
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
    acc.append((d, ghz, c["GRBM_GUI_ACTIVE"] / 8, busy))
acc = acc[len(acc) // 2:]   # second half of the dispatches: clocks settled
m = lambda k: sum(a[k] for a in acc) / len(acc)
print(f"{kern}: n={len(acc)}  {m(0)/1e3:7.1f} us  clock {m(1):.3f} GHz  cycles {m(2)/1e3:7.1f}K  mfma busy {100*m(3):.1f} %")
PY
