#!/bin/bash
# usage: _dbg/ab.sh "<workloads>" lib1 lib2 ...   (alternates the libraries twice)
W="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    RT_AMD_LIBRARY=$PWD/$lib timeout -k 10 300 python -u tools/quick_probe.py $W 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
