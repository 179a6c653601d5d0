#!/bin/bash
# Builds the measurement helpers (tools/*.c, tools/*.hip) into ${DPX_TOOLS_BIN:-/tmp/dpx_tools} — outside the repository, so that
# the tree that travels to a GPU box holds the product library and nothing else.  Run it ON the GPU box (hipcc is there).
set -e
BIN=${DPX_TOOLS_BIN:-/tmp/dpx_tools}
mkdir -p $BIN
gcc -O2 -o $BIN/pipe_source tools/pipe_source.c
gcc -O2 -o $BIN/pipe_sink tools/pipe_sink.c
gcc -O2 -Iinclude -o $BIN/block_async_bench tools/block_async_bench.c -Ldoppler_amd/lib -ldoppler_hip -Wl,-rpath,$PWD/doppler_amd/lib -Wl,-rpath,/opt/rocm/lib
for t in pcie_probe bar_probe membench rowbench; do
  hipcc --offload-arch=gfx950 -O2 tools/$t.hip -o $BIN/$t
done
echo "built into $BIN: $(ls $BIN | tr '\n' ' ')"
