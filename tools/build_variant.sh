#!/bin/bash
# Build a variant of the library with extra -D flags for A/B timing on one box:  tools/build_variant.sh NAME -DFLAG ...
# -> tools/bin/libdoppler_hip_NAME.so   (tools/ab_libs.sh copies it over doppler_amd/lib/libdoppler_hip.so ON THE GPU BOX between runs)
# Kernels and planner are both rebuilt with the flags (they share dpx_types.h); the API object is the shipped one.
set -e
NAME=$1; shift
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Iinclude \
    -mllvm -amdgpu-kernarg-preload-count=16 "$@" -c doppler_amd/csrc/dpx_kernels.hip -o /tmp/variant_$NAME.o 2>&1 | grep -v "argument unused" || true
g++ -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall "$@" -c doppler_amd/csrc/dpx_planner.cpp -o /tmp/variant_${NAME}_planner.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libdoppler_hip_$NAME.so /tmp/variant_$NAME.o doppler_amd/lib/dpx_api.o \
    /tmp/variant_${NAME}_planner.o doppler_amd/lib/orbit.o doppler_amd/lib/schedule.o -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libdoppler_hip.so
ls -la tools/bin/libdoppler_hip_$NAME.so
