#!/bin/bash
# Instruction-cache and issue counters of one bench run (a pass of its own: counters do not split over passes).
# usage: tools/pmc_icache.sh <outdir> [bench args...]      (RT_AMD_LIBRARY picks the build)
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="${*:---steps 1 --warmup 1 --spp 64 --no-cpu-baseline}"
run() {
  local name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM &&
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SMEM
python3 tools/pmc_summary.py "$OUT"
