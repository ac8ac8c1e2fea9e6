// pgcn_spmm_core.hip -- LDS-tiled SpMM for the dense core of a (degree-sorted) adjacency
// block, plus the list-driven fix-up kernel, for gfx950.
//
// Why: in the gather kernel (pgcn_spmm.hip) every stored entry pulls a 512 B feature row
// through L1 from L2; that path tops out at ~18-24 TB/s on MI355X.  Power-law graphs have a
// dense core: with vertices relabelled by decreasing degree, the tiles near the top-left
// of the matrix hold most of the entries (reddit-shaped R-MAT: the 128 x 128 tiles with
// >= 5 % fill hold 65 % of all entries).  For those tiles a workgroup stages the TC
// feature rows of a column panel ONCE into LDS (a contiguous 64 KB copy) and serves
// every entry of its 128 rows from LDS -- ds_read_b128 delivers ~4x the L1 rate -- so
// the L2 traffic of the core drops by the tile fill factor (25-50x).
//
// Work item ("piece") = one row tile (TR = 128 rows) x a run of dense column panels
// (TC = 128 columns each).  512 threads = 16 groups of 32 lanes; group g owns rows
// g, g+16, ... of the tile (8 rows, accumulators in registers, statically unrolled), lane
// s of a group owns features 4s..4s+3.  Per panel: stage -> barrier -> every group walks
// its 8 row segments.  (col-in-panel, val) pairs are read coalesced, parked in 512 B of
// LDS per wave and broadcast with ds_read_b128 (two pairs per read) -- ds_bpermute
// halves the throughput here because the LDS pipe is the bottleneck.  A piece writes its
// 128 partial rows to private slots; pgcn_spmm_fixup_f32 adds up, per output row, the
// slots listed for it (core pieces + gather-kernel tasks) in a fixed order: deterministic,
// no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pgcn_spmm_bodies.h"
#include "pgcn_once.h"

namespace {

using namespace pgcn_bodies;

// work: int4 {tile row index, first dense tile, one-past-last dense tile, first slot}; the body lives in
// pgcn_spmm_bodies.h (core_piece_body).
template <int VEC>
__global__ __launch_bounds__(kCoreThreads, 4) void spmm_core_kernel(
    const int4 *__restrict__ work, const int32_t *__restrict__ tile_panel,
    const int64_t *__restrict__ tile_base, const int32_t *__restrict__ seg_off,
    const int32_t *__restrict__ ccol, const float *__restrict__ cval, const float *__restrict__ B,
    int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pgcn_bodies::core_piece_body<VEC>(work[blockIdx.x], tile_panel, tile_base, seg_off, ccol, cval, B, ldb, ncols, f,
                                      partial, smem, blockIdx.y * 32 * VEC);
}

// fix: int4 {row, begin, count, 0}.  slot_ids != NULL: the slots of the row are
// slot_ids[begin .. begin+count); else they are begin .. begin+count.  Summed in list order.
template <int LPR, int VEC>
__global__ __launch_bounds__(256) void spmm_fixup_list_kernel(
    const int4 *__restrict__ fix, int64_t nfix, const int32_t *__restrict__ slot_ids,
    const int32_t *__restrict__ row_map, const float *__restrict__ partial, float *__restrict__ C,
    int64_t ldc, int32_t f, uint32_t flags) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane / LPR;
    const int sub = lane % LPR;
    const int64_t id = ((int64_t)blockIdx.x * 4 + wave) * G + grp;
    const int fcol = (blockIdx.y * LPR + sub) * VEC;
    if (id >= nfix || fcol >= f) return;
    const int4 t = fix[id];
    const int64_t orow = row_map ? row_map[t.x] : t.x;
    float *c = C + orow * ldc + fcol;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (flags & PGCN_SPMM_ACCUMULATE) vload<VEC>(acc, c);
    const float *p = partial + fcol;
    int s = 0;
    for (; s + 4 <= t.z; s += 4) {   // four independent loads in flight, summed in list order
        int64_t i0, i1, i2, i3;
        if (slot_ids) {
            i0 = slot_ids[t.y + s]; i1 = slot_ids[t.y + s + 1]; i2 = slot_ids[t.y + s + 2]; i3 = slot_ids[t.y + s + 3];
        } else {
            i0 = t.y + s; i1 = i0 + 1; i2 = i0 + 2; i3 = i0 + 3;
        }
        float x0[VEC], x1[VEC], x2[VEC], x3[VEC];
        vload<VEC>(x0, p + i0 * f);
        vload<VEC>(x1, p + i1 * f);
        vload<VEC>(x2, p + i2 * f);
        vload<VEC>(x3, p + i3 * f);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = (((acc[v] + x0[v]) + x1[v]) + x2[v]) + x3[v];
    }
    for (; s < t.z; ++s) {
        const int64_t i = slot_ids ? (int64_t)slot_ids[t.y + s] : (int64_t)(t.y + s);
        float x[VEC];
        vload<VEC>(x, p + i * f);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += x[v];
    }
    vstore<VEC>(c, acc);
}

bool aligned16(const void *a, const void *b, int64_t lda, int64_t ldb_, int32_t f) {
    return ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && lda % 4 == 0 && ldb_ % 4 == 0 && f % 4 == 0;
}

template <int LPR, int VEC>
int launch_fixup(const int32_t *fix, int64_t nfix, const int32_t *slot_ids, const int32_t *row_map,
                 const float *partial, float *C, int64_t ldc, int32_t f, uint32_t flags, hipStream_t s) {
    constexpr int G = 64 / LPR;
    const int ntiles = (f + LPR * VEC - 1) / (LPR * VEC);
    const int64_t per_block = 4 * G;
    const int64_t grid = (nfix + per_block - 1) / per_block;
    hipLaunchKernelGGL((spmm_fixup_list_kernel<LPR, VEC>), dim3((unsigned)grid, ntiles), dim3(256), 0, s,
                       reinterpret_cast<const int4 *>(fix), nfix, slot_ids, row_map, partial, C, ldc, f, flags);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

}  // namespace

extern "C" int pgcn_spmm_core_f32(const int32_t *work, int64_t nwork, const int32_t *tile_panel,
                                  const int64_t *tile_base, const int32_t *seg_off,
                                  const int32_t *ccol, const float *cval, const float *B,
                                  int64_t ldb, int64_t ncols, int32_t f, float *partial_ws,
                                  int64_t partial_ws_elems, int64_t nslots_total,
                                  pgcn_stream_t stream) {
    if (nwork < 0 || f <= 0 || ldb < f || ncols < 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_core_f32: bad sizes");
    if (nwork == 0) return PGCN_OK;
    if (!work || !tile_panel || !tile_base || !seg_off || !ccol || !cval || !B || !partial_ws)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_core_f32: null pointer");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_core_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_core_f32: too many pieces");
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = aligned16(B, partial_ws, ldb, 4, f);
    if (v4) {
        size_t smem = (size_t)(TC + 1) * 32 * 4 * 4 + (kCoreThreads / 64) * 512;
        static PgcnPerDeviceOnce once;
        if (int rc = once.run([&]() -> int {
                PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_core_kernel<4>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                return PGCN_OK;
            }))
            return rc;
        const int ntiles = (f + 127) / 128;
        hipLaunchKernelGGL((spmm_core_kernel<4>), dim3((unsigned)nwork, ntiles), dim3(kCoreThreads), smem, s,
                           reinterpret_cast<const int4 *>(work), tile_panel, tile_base, seg_off, ccol, cval,
                           B, ldb, ncols, f, partial_ws);
    } else {
        const size_t smem = (size_t)(TC + 1) * 32 * 4 + (kCoreThreads / 64) * 512;
        const int ntiles = (f + 31) / 32;
        hipLaunchKernelGGL((spmm_core_kernel<1>), dim3((unsigned)nwork, ntiles), dim3(kCoreThreads), smem, s,
                           reinterpret_cast<const int4 *>(work), tile_panel, tile_base, seg_off, ccol, cval,
                           B, ldb, ncols, f, partial_ws);
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

extern "C" int pgcn_spmm_fixup_f32(const int32_t *fix, int64_t nfix, const int32_t *slot_ids,
                                   const int32_t *row_map, const float *partial_ws, float *C,
                                   int64_t ldc, int32_t f, uint32_t flags, pgcn_stream_t stream) {
    if (nfix < 0 || f <= 0 || ldc < f) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_fixup_f32: bad sizes");
    if (nfix == 0) return PGCN_OK;
    if (!fix || !partial_ws || !C) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_fixup_f32: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = aligned16(C, partial_ws, ldc, 4, f);
    const int vec = v4 ? 4 : 1;
    const int nv = (f + vec - 1) / vec;
    int lpr = 1;
    while (lpr < nv && lpr < 64) lpr *= 2;
#define PGCN_FX(L, V) \
    if (lpr == L && vec == V) return launch_fixup<L, V>(fix, nfix, slot_ids, row_map, partial_ws, C, ldc, f, flags, s);
    PGCN_FX(1, 4) PGCN_FX(2, 4) PGCN_FX(4, 4) PGCN_FX(8, 4) PGCN_FX(16, 4) PGCN_FX(32, 4) PGCN_FX(64, 4)
    PGCN_FX(1, 1) PGCN_FX(2, 1) PGCN_FX(4, 1) PGCN_FX(8, 1) PGCN_FX(16, 1) PGCN_FX(32, 1) PGCN_FX(64, 1)
#undef PGCN_FX
    return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_fixup_f32: no kernel shape");
}
