#!/bin/bash
# HBM-side traffic and L2 hit rate of a bench.py run, counter passes of their own.
# usage: tools/pmc_traffic.sh <outdir> [bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="${*:---steps 1 --warmup 1 --spp 64 --no-cpu-baseline}"
run() {
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
run tcc2 FETCH_SIZE &&
run tcc3 WRITE_SIZE
python3 tools/pmc_summary.py "$OUT"
