#!/bin/bash
# Builds tools/micro/dense3_bench.bin against the in-tree libpgcn_hip.so (run __graft_entry__.build() first).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRC="$HERE/../../$PKG/csrc/pgcn_spmm_dense3.hip"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DNDEBUG -fno-slp-vectorize -Wno-unused-value"
objs=()
for v in 1 5; do
  "$HIPCC" $F -DPGCN_DENSE3_PROBE=$v -Dpgcn_spmm_dense_bf16x3_f32=pgcn_spmm_dense_bf16x3_probe${v}_f32 \
    -Dpgcn_dense_bf16x3_image_bytes=pgcn_dense_bf16x3_image_bytes_probe$v -c "$SRC" -o "$HERE/dense3_probe$v.o" &
  objs+=("$HERE/dense3_probe$v.o")
done
wait
"$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value -c "$HERE/dense3_bench.cpp" -o "$HERE/dense3_bench.o"
"$HIPCC" --offload-arch=gfx950 "$HERE/dense3_bench.o" "${objs[@]}" -o "$HERE/dense3_bench.bin" \
  -L"$HERE/../../$PKG/lib" -lpgcn_hip -Wl,-rpath,'$ORIGIN/../../'"$PKG/lib"
echo "built $HERE/dense3_bench.bin"
