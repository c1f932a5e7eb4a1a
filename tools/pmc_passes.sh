#!/bin/bash
# PMC passes over a short run of the bench workload; one rocprofv3 run per counter set.
# usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="${*:---steps 1 --warmup 1 --spp 64 --no-cpu-baseline}"
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o pmc -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM &&
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD &&
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum &&
run tcc2 FETCH_SIZE TCC_EA0_RDREQ_sum &&
run tcc3 WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
# vector memory path (the BVH workloads live here): L1 accesses, stalls, latency, address translation
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum &&
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN2_sum &&
run tcp3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TA_TCP_STATE_READ_sum &&
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
