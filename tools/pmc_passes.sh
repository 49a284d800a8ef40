#!/bin/bash
# Hardware-counter passes over one program, one rocprofv3 run per counter group (a run holds at most
# 8 SQ, 4 TCP, 4 TCC, 2 TA counters).  Usage:
#   tools/pmc_passes.sh OUTDIR KERNEL_SUBSTRING -- program args...
# Prints, per counter, the average over the dispatches whose kernel name contains KERNEL_SUBSTRING.
set -u
out=$1; kern=$2; shift 3
export TMPDIR=/tmp
groups=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"
  "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"
  "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
)
i=0
for g in "${groups[@]}"; do
  d=$out/pass$i
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $d -o p -- "$@" > $d.log 2>&1 || { echo "pass $i failed"; tail -3 $d.log; exit 1; }
  i=$((i+1))
done
python3 - "$out" "$kern" <<'EOF'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:42s} avg {sum(v)/len(v):16.0f}  n={len(v)}")
EOF
