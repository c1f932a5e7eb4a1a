#!/bin/bash
# Counter passes of every workload on the shipped binary, then the digest bench.py reads.
# usage: tools/round_pmc.sh <outdir> <tag>   (run from the repository root on the GPU box)
# Each workload runs one warm-up and one timed frame per pass, so a pass sees twice
# config.segments_per_step of its bench line; the listings go to <outdir>/<tag>_pmc_<name>.txt.
set -u
OUT=$1; TAG=$2
mkdir -p "$OUT"
one() {
  local name=$1; shift
  bash tools/pmc_quick.sh "$OUT/pmc_$name" "$@" > "$OUT/${TAG}_pmc_$name.txt" 2>&1 || { echo "$name failed"; tail -5 "$OUT/${TAG}_pmc_$name.txt"; return 1; }
  python3 - "$OUT/pmc_$name/sq1.log" "$name" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line)
        print(sys.argv[2], "segments in a pass:", 2 * j["config"]["segments_per_step"], "kernel:", j["roofline"].get("kernel"))
PY
}
one c3_spp64 --steps 1 --warmup 1 --spp 64 --no-cpu-baseline &&
one c1_spp400 --workload c1 --steps 1 --warmup 1 --spp 400 --no-cpu-baseline &&
one c2_spp36 --workload c2 --steps 1 --warmup 1 --spp 36 --no-cpu-baseline &&
one c4_spp64 --workload c4 --steps 1 --warmup 1 --spp 64 --no-cpu-baseline &&
one c5_spp1 --workload c5 --steps 1 --warmup 1 --spp 1 --no-cpu-baseline
