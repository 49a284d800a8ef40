cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# usage: conv_w8_split.sh ["B Cin H Cout k s p dtype" ...]   (default: five ResNet-18 layers at batch 256)
if [ $# -eq 0 ]; then set -- "256 64 56 64 3 1 1 f32" "256 64 56 64 3 1 1 bf16" "256 128 28 128 3 1 1 f32" "256 3 224 64 7 2 3 f32" "256 512 7 512 3 1 1 f32"; fi
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pw8_$tag -o t -- python3 $R/tools/conv_w8_probe.py $cfg 20 > /dev/null 2>&1 || exit 1
  echo "== $cfg"; 
  f=$(find $R/gpurun_out/pw8_$tag -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('conv_w8','gemm_w8f')):
        print(f"  {n[:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
