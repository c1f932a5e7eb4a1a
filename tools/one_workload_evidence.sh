#!/bin/bash
# Bench line and counter passes of ONE workload on the shipped binary (what tools/round_evidence.sh and
# tools/round_pmc.sh do for all of them).  usage: tools/one_workload_evidence.sh <outdir> <tag> <name> <bench args> -- <pmc bench args>
set -u
OUT=$1; TAG=$2; NAME=$3; shift 3
BENCH=(); while [ "$1" != "--" ]; do BENCH+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout -k 10 900 python3 bench.py "${BENCH[@]}" > "$OUT/${TAG}_bench_${NAME}.json" 2> "$OUT/${NAME}.err" || exit 1
echo "$NAME bench done"
bash tools/pmc_quick.sh "$OUT/pmc_$NAME" "$@" > "$OUT/${TAG}_pmc_${NAME}.txt" 2>&1 || exit 1
grep "^{" "$OUT/pmc_$NAME/sq1.log" | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print('segments in a pass:', 2*j['config']['segments_per_step'])"
