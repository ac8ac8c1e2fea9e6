#!/bin/bash
# Builds tools/micro/dense_fused_bench.bin: the kernels of gemm/pgcn_dense.hip linked in directly (no rocBLAS, no Python).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRC="$HERE/../../$PKG/gemm/pgcn_dense.hip"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
"$HIPCC" $F -c "$SRC" -o "$HERE/dense_fused_kernels.o" ${PGCN_EXTRA_FLAGS:-} &
objs=("$HERE/dense_fused_kernels.o")
# probe builds under their own symbol names: p0 = pipelined steps, loads after the stores, masked operand by whole tiles;
# p2 = the first version (LDS reads / wait / MFMAs per step, cur = nxt copies, no prefetch, whole tiles)
declare -A VAR=([0]="-DPGCN_DENSE_PIPE=1 -DPGCN_DENSE_PREFETCH=0 -DPGCN_DENSE_MASK_PIPE=0" [2]="-DPGCN_DENSE_PIPE=0 -DPGCN_DENSE_PREFETCH=0 -DPGCN_DENSE_MASK_PIPE=0")
for v in 0 2; do
  "$HIPCC" $F ${VAR[$v]} -Dpgcn_dense=pgcn_dense_p$v -Dpgcn_linear_relu_f32=pgcn_linear_relu_f32_p$v \
    -Dpgcn_linear_relu_grad_input_f32=pgcn_linear_relu_grad_input_f32_p$v -Dpgcn_dense_last_error=pgcn_dense_last_error_p$v \
    -c "$SRC" -o "$HERE/dense_fused_kernels_p$v.o" &
  objs+=("$HERE/dense_fused_kernels_p$v.o")
done
wait
"$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 -c "$HERE/dense_fused_bench.cpp" -o "$HERE/dense_fused_bench.o"
"$HIPCC" --offload-arch=gfx950 "$HERE/dense_fused_bench.o" "${objs[@]}" -o "$HERE/dense_fused_bench.bin"
echo "built $HERE/dense_fused_bench.bin"
