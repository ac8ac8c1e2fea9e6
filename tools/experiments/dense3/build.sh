#!/bin/bash
# Builds tools/experiments/dense3/dense3_bench.bin: the harness + the kernel (as it is, two timing-only probe builds
# and one with per-wave phase timers), linked against the in-tree libpgcn_hip.so (fp32-MFMA kernel to compare with, pgcn_set_error); run
# __graft_entry__.build() first.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
SRC="$HERE/pgcn_spmm_dense3.hip"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DNDEBUG -Wno-unused-value"
"$HIPCC" $F -c "$SRC" -o "$HERE/dense3.o"
for v in 1 2 3; do
  "$HIPCC" $F -DPGCN_DENSE3_PROBE=$v -Dpgcn_spmm_dense_bf16x3_f32=pgcn_spmm_dense_bf16x3_probe${v}_f32 -c "$SRC" -o "$HERE/dense3_probe$v.o"
done
"$HIPCC" $F -c "$HERE/dense3_bench.cpp" -o "$HERE/dense3_bench.o"
"$HIPCC" --offload-arch=gfx950 "$HERE/dense3_bench.o" "$HERE/dense3.o" "$HERE/dense3_probe1.o" "$HERE/dense3_probe2.o" "$HERE/dense3_probe3.o" -o "$HERE/dense3_bench.bin" \
  -L"$HERE/../../../$PKG/lib" -lpgcn_hip -Wl,-rpath,'$ORIGIN/../../../'"$PKG/lib"
echo "built $HERE/dense3_bench.bin"
