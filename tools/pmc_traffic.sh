#!/bin/bash
# usage: pmc2.sh OUTDIR KERNEL -- program args
out=$1; kern=$2; shift 3
export TMPDIR=/tmp
mkdir -p $out
i=0
for g in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  d=$out/pass$i
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $d -o p -- "$@" > $d.log 2>&1 || { echo "pass $i failed"; tail -3 $d.log; exit 1; }
  i=$((i+1))
done
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:30s} avg {sum(v)/len(v):16.0f}  n={len(v)}")
PY
