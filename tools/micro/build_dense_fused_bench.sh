#!/bin/bash
# Builds tools/micro/dense_fused_bench.bin: the kernels of gemm/pgcn_dense.hip linked in directly (no rocBLAS, no Python).
# PGCN_EXTRA_FLAGS reaches the kernel file (an A/B build: compile twice under different output names).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$HERE/../../$PKG/gemm/pgcn_dense.hip" -o "$HERE/dense_fused_kernels.o" ${PGCN_EXTRA_FLAGS:-}
"$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 -c "$HERE/dense_fused_bench.cpp" -o "$HERE/dense_fused_bench.o"
"$HIPCC" --offload-arch=gfx950 "$HERE/dense_fused_bench.o" "$HERE/dense_fused_kernels.o" -o "$HERE/dense_fused_bench.bin"
echo "built $HERE/dense_fused_bench.bin"
