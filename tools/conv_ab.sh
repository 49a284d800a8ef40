#!/bin/bash
# Kernel-trace A/B of the 3 x 3 planar conv kernel against the im2col-order kernel on one layer:
#   tools/conv_ab.sh OUTDIR B Cin H Cout [out dtype]     (run from /tmp on the GPU box)
out=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in 1 0; do
  MIQ_CONV_PLANAR=$mode timeout -k 10 90 rocprofv3 --kernel-trace -d $out/m$mode -o p -- python3 $R/tools/conv_probe.py $1 $2 $3 $4 3 1 1 100 ${5:-i32} > /dev/null 2>&1 || exit 1
  echo "MIQ_CONV_PLANAR=$mode  B=$1 $2->$4 ${3}x${3} out=${5:-i32}"
  python3 $R/tools/prof_kernels.py $(find $out/m$mode -name "*.db" | head -1) conv
done
