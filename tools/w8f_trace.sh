#!/bin/bash
# Kernel-trace durations of miq_gemm_w8's launches for a few (M, dtype) cases on [14336,4096] (run on the GPU box).
#   tools/w8f_trace.sh OUTDIR
out=$1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for case in "16 f32" "16 bf16" "64 f32" "64 bf16" "128 f32" "128 bf16" "256 f32" "256 bf16" "2048 bf16" "2048 f32"; do
  set -- $case
  timeout -k 10 90 rocprofv3 --kernel-trace -d $out/m$1_$2 -o p -- python3 $R/tools/w8f_probe.py $1 14336 4096 $2 40 > /dev/null 2>&1 || exit 1
  echo "M=$1 $2 [14336,4096]"
  python3 $R/tools/prof_kernels.py $(find $out/m$1_$2 -name "*.db" | head -1) w8f
  python3 $R/tools/prof_kernels.py $(find $out/m$1_$2 -name "*.db" | head -1) split_planes
  rm -rf $out/m$1_$2
done
