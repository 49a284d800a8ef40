#!/bin/bash
# Counter passes over one program, one rocprofv3 run per group; prints per kernel-name substring the average of every
# counter over that kernel's dispatches.  Usage: tools/pmc_by_kernel.sh OUTDIR "SUBSTR1 SUBSTR2" -- python3 prog args...
set -u
out=$1; kerns=$2; shift 3
export TMPDIR=/tmp
mkdir -p $out
groups=(
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"
  "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
)
i=0
for g in "${groups[@]}"; do
  d=$out/pass$i
  timeout -s KILL 150 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -o p -- "$@" > $d.log 2>&1 || { echo "pass $i failed"; tail -5 $d.log; exit 1; }
  i=$((i+1))
done
python3 - "$out" "$kerns" <<'PY'
import csv, glob, sys, collections
out, kerns = sys.argv[1], sys.argv[2].split()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in kerns:
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        v = v[len(v)//2:]   # the second half of the dispatches (clocks ramped)
        print(f"{k:14s} {c:36s} avg {sum(v)/len(v):16.0f}  n={len(v)}")
for f in sorted(glob.glob(out + "/pass0/**/*kernel_trace.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in kerns:
            if k in r["Kernel_Name"]:
                acc[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in acc.items():
        v = v[len(v)//2:]
        print(f"{k:14s} duration_ns avg {sum(v)/len(v):12.0f} n={len(v)}")
PY
