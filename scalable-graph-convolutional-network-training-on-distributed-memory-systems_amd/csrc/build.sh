#!/bin/bash
# Builds libpgcn_hip.so for gfx950 in-tree (../lib/).  hipcc cross-compiles without a GPU.
# The library links against libamdhip64.so.7 / librccl.so.1 by SONAME only: inside a
# PyTorch process the loader re-uses the copies torch already mapped (same SONAMEs),
# so torch's streams / device pointers are valid in here; stand-alone it uses /opt/rocm.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
ROCM="${ROCM_PATH:-/opt/rocm}"
HIPCC="${HIPCC:-$ROCM/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast -DNDEBUG ${PGCN_EXTRA_FLAGS:-}"
pids=()
for src in pgcn_spmm.hip pgcn_spmm_core.hip pgcn_spmm_dense3.hip pgcn_spmm_strip.hip pgcn_spmm_heads.hip pgcn_gat_blocks.hip pgcn_loss.hip pgcn_rows.hip pgcn_gat.hip; do
  # (packed fp32 VALU beside MFMAs costs the matrix pipe ~12 cycles per instruction: no SLP packing of the A split)
  extra=""; { [ "$src" = pgcn_spmm_dense3.hip ] || [ "$src" = pgcn_gat_blocks.hip ]; } && extra="-fno-slp-vectorize"
  "$HIPCC" $FLAGS $extra -c "$HERE/$src" -o "$OUT/${src%.hip}.o" &
  pids+=($!)
done
"$HIPCC" $FLAGS -c "$HERE/pgcn_core.cpp" -o "$OUT/pgcn_core.o" & pids+=($!)
"$HIPCC" $FLAGS -c "$HERE/pgcn_exchange.cpp" -o "$OUT/pgcn_exchange.o" & pids+=($!)
"$HIPCC" $FLAGS -c "$HERE/pgcn_mtx.cpp" -o "$OUT/pgcn_mtx.o" & pids+=($!)
"$HIPCC" $FLAGS -c "$HERE/pgcn_maps.cpp" -o "$OUT/pgcn_maps.o" & pids+=($!)
"$HIPCC" $FLAGS -c "$HERE/pgcn_shard.cpp" -o "$OUT/pgcn_shard.o" & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libpgcn_hip.so" \
  "$OUT/pgcn_spmm.o" "$OUT/pgcn_spmm_core.o" "$OUT/pgcn_spmm_dense3.o" "$OUT/pgcn_spmm_strip.o" "$OUT/pgcn_spmm_heads.o" "$OUT/pgcn_gat_blocks.o" "$OUT/pgcn_loss.o" "$OUT/pgcn_rows.o" "$OUT/pgcn_gat.o" "$OUT/pgcn_core.o" "$OUT/pgcn_exchange.o" "$OUT/pgcn_mtx.o" "$OUT/pgcn_maps.o" "$OUT/pgcn_shard.o" \
  -L"$ROCM/lib" -lrccl -lpthread
echo "built $OUT/libpgcn_hip.so"
