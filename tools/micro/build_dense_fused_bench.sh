#!/bin/bash
# Builds tools/micro/dense_fused_bench.bin (binds lib/libpgcn_gemm.so with dlopen: build the package first) and, for the A/B,
#   tools/micro/libpgcn_dense_r05.so    the r05 kernels (tools/micro/r05/: the sources as they were at the end of round 5),
#   tools/micro/libpgcn_wgrad_probe.so  gemm/pgcn_wgrad.hip with the fp32-MFMA probe entry point.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$HERE/r05" "$HERE/r05/pgcn_dense.hip" -o "$HERE/libpgcn_dense_r05.so" &
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -DPGCN_WGRAD_PROBES "$HERE/../../$PKG/gemm/pgcn_wgrad.hip" -o "$HERE/libpgcn_wgrad_probe.so" &
"$HIPCC" --offload-arch=gfx950 -O2 -std=c++17 "$HERE/dense_fused_bench.cpp" -o "$HERE/dense_fused_bench.bin" -ldl &
wait
echo "built $HERE/dense_fused_bench.bin"
