#!/bin/bash
# Segment rates of several builds of the library on one box, in turn, twice over (boxes drift):
# usage: tools/ab_probe.sh "<workload[:spp] ...>" <variant> [<variant> ...]    ("shipped" = librt_amd.so)
set -e
cd "$(dirname "$0")/.."
W=$1; shift
for round in 1 2; do
  for v in "$@"; do
    lib=$PWD/raytracing_amd/librt_amd_$v.so
    [ "$v" = shipped ] && lib=$PWD/raytracing_amd/librt_amd.so
    echo "== $v (round $round)"
    RT_AMD_LIBRARY=$lib timeout -k 10 240 python3 tools/quick_probe.py $W
  done
done
