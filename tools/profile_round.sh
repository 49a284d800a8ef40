#!/bin/bash
# The measurements the committed files under profiles/ come from, in one GPU call:
#   tools/profile_round.sh OUTDIR
#   1. bench.py (default flags)                         -> OUTDIR/bench.json, OUTDIR/bench_extra.json (the long tables)
#   2. rocprofv3 --kernel-trace --stats over bench.py   -> OUTDIR/stats/*kernel_stats.csv (headline step), OUTDIR/stats_extras/ (with the extras)
#   3. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE     -> OUTDIR/fetch, OUTDIR/write (separate runs)
#   4. tools/kbench.py                                  -> OUTDIR/kbench.txt
# Every GPU step runs under its own time limit and the script stops at the first failure.
set -u
out=$1
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
tail -1 "$out/bench.json"
cp gpurun_out/bench_extra.json "$out/bench_extra.json" 2>/dev/null
# (the headline step alone, so that the GEMM kernel's average duration in the summary is that of the 8192^3 launch; then once with the extras)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 bench.py --no-cpu-baseline --no-extras > "$out/stats.log" 2>&1 || { echo "stats failed"; tail -5 "$out/stats.log"; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_extras" -o bench -- python3 bench.py --no-cpu-baseline --steps 20 > "$out/stats_extras.log" 2>&1 || { echo "stats (extras) failed"; tail -5 "$out/stats_extras.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > "$out/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$out/fetch.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o p -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 > "$out/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$out/write.log"; exit 1; }
timeout -k 10 300 python3 tools/kbench.py > "$out/kbench.txt" 2>&1 || { echo "kbench failed"; tail -5 "$out/kbench.txt"; exit 1; }
python3 - "$out" <<'EOF'
import csv, glob, json, sys, collections, statistics
out = sys.argv[1]
res = {}
for which, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{which}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == key:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for short in ("quantize_rows_reg_kernel", "gemm_i8_256_kernel"):
            if short in k:
                res.setdefault(short, {})[key] = statistics.median(v)   # the bench shape; stray launches of other shapes drop out
for short, d in list(res.items()):
    f, w = d.get("FETCH_SIZE", 0.0) * 2.0, d.get("WRITE_SIZE", 0.0)   # gfx950: FETCH_SIZE counts 64 B per 128-B request
    d.update({"FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
              "hbm_bytes_per_launch": int((f + w) * 1024)})
    d.pop("FETCH_SIZE", None); d.pop("WRITE_SIZE", None)
sys.path.insert(0, ".")
import bench
res["source_sha"] = bench.kernel_source_sha()   # bench.py reports these numbers only for exactly these kernel sources
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res))
EOF
