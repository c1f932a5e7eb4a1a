#!/bin/bash
# End-of-round evidence: bench lines of every workload, kernel trace of the headline bench, counter
# passes.  usage: tools/round_evidence.sh <outdir> <tag>   (run from the repository root on the GPU box)
set -u
OUT=$1; TAG=$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 2 > "$OUT/${TAG}_bench_c3.json" 2> "$OUT/${TAG}_bench_c3.err" || exit 1
echo "c3 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace_c3" -o trace -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_c3_under_rocprof.json" 2> "$OUT/${TAG}_trace_c3.err" || exit 1
cp "$OUT/${TAG}_trace_c3/trace_kernel_stats.csv" "$OUT/${TAG}_bench_c3_kernel_stats.csv" 2>/dev/null
echo "c3 trace done"
timeout -k 10 300 python3 bench.py --workload c1 --steps 50 --warmup 5 > "$OUT/${TAG}_bench_c1_as_shipped.json" 2> "$OUT/${TAG}_bench_c1.err" || exit 1
echo "c1 done"
timeout -k 10 400 python3 bench.py --workload c2 --steps 10 --warmup 2 > "$OUT/${TAG}_bench_c2_full.json" 2> "$OUT/${TAG}_bench_c2.err" || exit 1
echo "c2 done"
timeout -k 10 600 python3 bench.py --workload c4 --steps 3 --warmup 1 > "$OUT/${TAG}_bench_c4_full.json" 2> "$OUT/${TAG}_bench_c4.err" || exit 1
echo "c4 done"
timeout -k 10 300 python3 bench.py --wavefront --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_c3_wavefront.json" 2> "$OUT/${TAG}_bench_c3_wavefront.err" || exit 1
timeout -k 10 300 python3 bench.py --regroup off --steps 10 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_c3_lane_bound.json" 2> "$OUT/${TAG}_bench_c3_lane_bound.err" || exit 1
echo "variants done"
timeout -k 10 500 python3 bench.py --workload c5 --steps 1 --warmup 1 > "$OUT/${TAG}_bench_c5_full.json" 2> "$OUT/${TAG}_bench_c5.err" || exit 1
echo "c5 done"
