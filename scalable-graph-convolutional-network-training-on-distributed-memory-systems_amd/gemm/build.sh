#!/bin/bash
# Builds lib/libpgcn_gemm.so (host code only: rocBLAS calls by solution index).  Links librocblas.so.5 / libamdhip64 by
# SONAME: inside a PyTorch process the loader re-uses the copies PyTorch mapped.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"; mkdir -p "$OUT"
ROCM="${ROCM_PATH:-/opt/rocm}"
"${HIPCC:-$ROCM/bin/hipcc}" -O2 -std=c++17 -fPIC -shared -Wall -I"$ROCM/include" "$HERE/pgcn_gemm.cpp" \
  -o "$OUT/libpgcn_gemm.so" -L"$ROCM/lib" -lrocblas -lamdhip64 -lpthread
echo "built $OUT/libpgcn_gemm.so"
