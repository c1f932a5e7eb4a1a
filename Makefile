# Builds the product library (HIP kernels + C ABI + host scene layer) for gfx950 and the
# CPU oracle (test infrastructure only; the product never links it).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CXXSTD = -std=c++17
# fp32 / and sqrt as v_rcp_f32 / v_sqrt_f32 sequences (~1-2.5 ulp) instead of the correctly
# rounded expansions: the parity tolerances are 1e-5 relative, the kernels are VALU-bound
HIPFLAGS = --offload-arch=$(ARCH) -O3 $(CXXSTD) -fPIC -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt \
           -Wall -Wno-unused-function
HOSTFLAGS = -O2 $(CXXSTD) -fPIC -Wall -Wextra -Wno-unused-parameter -ffp-contract=off

LIB = raytracing_amd/librt_amd.so
HOST_SRC = raytracing_amd/host/scene.cpp raytracing_amd/host/flatten.cpp raytracing_amd/host/config.cpp \
           raytracing_amd/host/image_io.cpp raytracing_amd/host/host_capi.cpp raytracing_amd/host/render.cpp
HOST_OBJ = $(HOST_SRC:.cpp=.o)
DEV_OBJ = raytracing_amd/csrc/rt_device.o

# the reference's binary (main.rs) on this backend
MAIN = bin/rt_main

all: $(LIB) $(MAIN) oracle sanitize

$(DEV_OBJ): raytracing_amd/csrc/rt_device.hip raytracing_amd/csrc/rt_kernels.h raytracing_amd/csrc/rt_accel.h raytracing_amd/csrc/rt_math.h raytracing_amd/csrc/rt_leafword.h raytracing_amd/csrc/rt_flat.h include/rt_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

raytracing_amd/host/%.o: raytracing_amd/host/%.cpp raytracing_amd/host/raytracing.hpp include/rt_amd.h
	g++ $(HOSTFLAGS) -c $< -o $@

$(LIB): $(DEV_OBJ) $(HOST_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(DEV_OBJ) $(HOST_OBJ) -lz

$(MAIN): raytracing_amd/host/main.o $(LIB)
	mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -o $@ $< -Lraytracing_amd -lrt_amd -Wl,-rpath,'$$ORIGIN/../raytracing_amd'

# the host layer alone (no HIP) under AddressSanitizer + UBSan, with a fuzzing driver
SAN_SRC = raytracing_amd/host/scene.cpp raytracing_amd/host/flatten.cpp raytracing_amd/host/config.cpp \
          raytracing_amd/host/image_io.cpp raytracing_amd/host/host_capi.cpp tests/native/host_sanitize.cpp
sanitize: bin/host_sanitize
bin/host_sanitize: $(SAN_SRC) raytracing_amd/host/raytracing.hpp raytracing_amd/csrc/rt_accel.h raytracing_amd/csrc/rt_flat.h include/rt_amd.h
	mkdir -p bin
	g++ -O1 -g $(CXXSTD) -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
	    -Wall -Wextra -Wno-unused-parameter -ffp-contract=off -o $@ $(SAN_SRC) -lz

# the same library with per-step counters in the regrouping kernel (tools/regroup_steps.py)
STATS_LIB = raytracing_amd/librt_amd_stats.so
stats-lib: $(STATS_LIB)
$(STATS_LIB): raytracing_amd/csrc/rt_device.hip raytracing_amd/csrc/rt_kernels.h raytracing_amd/csrc/rt_accel.h raytracing_amd/csrc/rt_math.h raytracing_amd/csrc/rt_leafword.h raytracing_amd/csrc/rt_flat.h include/rt_amd.h $(HOST_OBJ)
	$(HIPCC) $(HIPFLAGS) -DRT_REGROUP_STATS -c $< -o raytracing_amd/csrc/rt_device_stats.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ raytracing_amd/csrc/rt_device_stats.o $(HOST_OBJ) -lz

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) $(STATS_LIB) raytracing_amd/csrc/rt_device_stats.o $(DEV_OBJ) $(HOST_OBJ) $(MAIN) raytracing_amd/host/main.o bin/host_sanitize
	$(MAKE) -C oracle clean

.PHONY: all oracle clean sanitize stats-lib
