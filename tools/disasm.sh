#!/bin/bash
# Disassemble the device code of the built library: tools/disasm.sh [out.s]   (default /tmp/dpx_dev.s)
set -e
LLVM=/opt/rocm/lib/llvm/bin
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/dpx_dev.s}
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $ROOT/doppler_amd/lib/libdoppler_hip.so $T/unused.so
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
$LLVM/llvm-objdump -d $T/dev.co > $OUT
rm -rf $T
echo $OUT
