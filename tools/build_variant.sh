#!/bin/bash
# Builds another copy of the library with extra compiler flags (kernel experiments):
#   tools/build_variant.sh <name> [-DRT_SHADE_QUORUM=32 ...]  ->  raytracing_amd/librt_amd_<name>.so
# Run anything against it with RT_AMD_LIBRARY=$PWD/raytracing_amd/librt_amd_<name>.so.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
make -s raytracing_amd/librt_amd.so >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-function "$@" -c raytracing_amd/csrc/rt_device.hip -o /tmp/rt_device_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o raytracing_amd/librt_amd_$name.so /tmp/rt_device_$name.o raytracing_amd/host/scene.o \
    raytracing_amd/host/flatten.o raytracing_amd/host/config.o raytracing_amd/host/image_io.o raytracing_amd/host/host_capi.o raytracing_amd/host/render.o -lz
echo built raytracing_amd/librt_amd_$name.so
