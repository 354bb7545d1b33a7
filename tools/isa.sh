#!/bin/bash
# A generated pipeline kernel -> gfx950 assembly, offline (no GPU): the options hiprtc gets (runtime.cpp), plus the HIP runtime header hiprtc
# includes by itself.  usage: tools/isa.sh KERNEL.hip OUT.s [-DRSQ_LAZY=1 ...]
set -e
src=$1; out=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$(dirname "$0")/../resql_amd/csrc/kernels" -include hip/hip_runtime.h "$@" \
    -S --cuda-device-only -o "$out" "$src"
grep -E "^\s+\.(sgpr|vgpr)_count|NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize" "$out" | sort | uniq | head -20
