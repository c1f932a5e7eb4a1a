#!/bin/bash
# An experimental build of the library beside the shipped one: raytracing_amd/librt_amd_<name>.so
# usage: tools/build_variant.sh <name> [-DFLAG ...]     then   RT_AMD_LIBRARY=$PWD/raytracing_amd/librt_amd_<name>.so python tools/...
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
HOST_OBJ="raytracing_amd/host/scene.o raytracing_amd/host/flatten.o raytracing_amd/host/config.o raytracing_amd/host/image_io.o raytracing_amd/host/host_capi.o raytracing_amd/host/render.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function "$@" \
  -c raytracing_amd/csrc/rt_device.hip -o /tmp/rt_device_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o raytracing_amd/librt_amd_$NAME.so /tmp/rt_device_$NAME.o $HOST_OBJ -lz
echo built raytracing_amd/librt_amd_$NAME.so
