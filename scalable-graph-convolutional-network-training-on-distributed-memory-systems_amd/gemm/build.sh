#!/bin/bash
# Builds lib/libpgcn_gemm.so: the dense products of a layer beside the graded aggregation path --
#   pgcn_gemm.cpp   host code, rocBLAS calls by solution index;
#   pgcn_dense.hip  relu(X.W^T) and its input gradient on the bf16 matrix cores (gfx950; index arithmetic: pgcn_dense_tile.h);
#   pgcn_wgrad.hip  the weight gradient Gm^T.X on the same cores.
# Links librocblas.so.5 / libamdhip64 by SONAME: inside a PyTorch process the loader re-uses the copies PyTorch mapped.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"; mkdir -p "$OUT"
ROCM="${ROCM_PATH:-/opt/rocm}"
HIPCC="${HIPCC:-$ROCM/bin/hipcc}"
# (-fno-slp-vectorize: packed fp32 VALU beside MFMAs costs the matrix pipe ~12 cycles per instruction -- the bf16 split stays scalar)
"$HIPCC" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -c "$HERE/pgcn_dense.hip" -o "$OUT/pgcn_dense.o" ${PGCN_EXTRA_FLAGS:-} &
"$HIPCC" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -c "$HERE/pgcn_wgrad.hip" -o "$OUT/pgcn_wgrad.o" ${PGCN_EXTRA_FLAGS:-} &
"$HIPCC" -O2 -std=c++17 -fPIC -Wall -I"$ROCM/include" -c "$HERE/pgcn_gemm.cpp" -o "$OUT/pgcn_gemm.o" &
wait
"$HIPCC" -shared -fPIC "$OUT/pgcn_gemm.o" "$OUT/pgcn_dense.o" "$OUT/pgcn_wgrad.o" -o "$OUT/libpgcn_gemm.so" -L"$ROCM/lib" -lrocblas -lamdhip64 -lpthread
echo "built $OUT/libpgcn_gemm.so"
