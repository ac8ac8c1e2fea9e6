#!/bin/bash
# Builds tools/micro/dense_fused_bench.bin: the kernels of gemm/pgcn_dense.hip linked in directly (no rocBLAS, no Python).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRC="$HERE/../../$PKG/gemm/pgcn_dense.hip"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
"$HIPCC" $F -c "$SRC" -o "$HERE/dense_fused_kernels.o" ${PGCN_EXTRA_FLAGS:-} &
"$HIPCC" $F -c "$HERE/../../$PKG/gemm/pgcn_wgrad.hip" -o "$HERE/dense_fused_wgrad.o" &
objs=("$HERE/dense_fused_kernels.o" "$HERE/dense_fused_wgrad.o")
# probe builds under their own symbol names: p0 = pipelined steps, loads after the stores, masked operand by whole tiles;
# p2 = the first version (LDS reads / wait / MFMAs per step, cur = nxt copies, no prefetch, whole tiles); f1 / f2 = the candidates
# of the end of r04 (predicate-free inner tiles; + non-temporal stores; c1 = + the transposed tile with 16-byte stores); t1-t3 = timing-only probes (see gemm/pgcn_dense.hip)
declare -A VAR=([p0]="-DPGCN_DENSE_PIPE=1 -DPGCN_DENSE_PREFETCH=0 -DPGCN_DENSE_MASK_PIPE=0"
                [p2]="-DPGCN_DENSE_PIPE=0 -DPGCN_DENSE_PREFETCH=0 -DPGCN_DENSE_MASK_PIPE=0"
                [f1]="-DPGCN_DENSE_FASTPATH=1" [f2]="-DPGCN_DENSE_FASTPATH=1 -DPGCN_DENSE_NT_STORE=1"
                [c1]="-DPGCN_DENSE_FASTPATH=1 -DPGCN_DENSE_CT=1"
                [f3]="-DPGCN_DENSE_FASTPATH=1 -DPGCN_DENSE_PREFETCH=0" [c2]="-DPGCN_DENSE_FASTPATH=1 -DPGCN_DENSE_CT=1 -DPGCN_DENSE_PREFETCH=0"
                [g1]="-DPGCN_DENSE_FASTPATH=1 -DPGCN_DENSE_CT=1 -DPGCN_DENSE_SPREAD=1"
                [t1]="-DPGCN_DENSE_PROBE=1" [t2]="-DPGCN_DENSE_PROBE=2" [t3]="-DPGCN_DENSE_PROBE=3")
for v in p0 p2 f1 f2 f3 c1 c2 g1 t1 t2 t3; do
  "$HIPCC" $F ${VAR[$v]} -Dpgcn_dense=pgcn_dense_$v -Dpgcn_linear_relu_f32=pgcn_linear_relu_f32_$v \
    -Dpgcn_linear_relu_grad_input_f32=pgcn_linear_relu_grad_input_f32_$v -Dpgcn_dense_last_error=pgcn_dense_last_error_$v \
    -c "$SRC" -o "$HERE/dense_fused_kernels_$v.o" &
  objs+=("$HERE/dense_fused_kernels_$v.o")
done
wait
"$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 -c "$HERE/dense_fused_bench.cpp" -o "$HERE/dense_fused_bench.o"
"$HIPCC" --offload-arch=gfx950 "$HERE/dense_fused_bench.o" "${objs[@]}" -o "$HERE/dense_fused_bench.bin"
echo "built $HERE/dense_fused_bench.bin"
