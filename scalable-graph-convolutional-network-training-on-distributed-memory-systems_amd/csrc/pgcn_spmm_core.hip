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

#include "pgcn_device.h"
#include "pgcn_internal.h"

namespace {

constexpr int TR = PGCN_CORE_TR;   // rows per tile
constexpr int TC = PGCN_CORE_TC;   // columns per panel
constexpr int kCoreThreads = 512;
constexpr int NG = kCoreThreads / 32;   // groups per workgroup
constexpr int RW = TR / NG;             // rows per group
#ifndef PGCN_CORE_BATCH
#define PGCN_CORE_BATCH 4
#endif
constexpr int CB = PGCN_CORE_BATCH;     // LDS row reads in flight per batch
#ifndef PGCN_CORE_STAGE_BATCH
#define PGCN_CORE_STAGE_BATCH 4
#endif
static_assert(TR % NG == 0, "tile rows must divide over the groups");

template <int VEC>
__device__ __forceinline__ void lds_row(float (&x)[VEC], const float *panel, int c, int sub) {
    vload<VEC>(x, panel + (c * 32 + sub) * VEC);
}

// work: int4 {tile row index, first dense tile, one-past-last dense tile, first slot}
//
// Latency structure: the (col,val) stream of a segment is the only HBM-latency-bound load in
// the loop, so (1) the first 32 pairs of ALL eight segments of a group are requested before
// the panel is staged (they land behind the staging copy and the barrier), (2) longer
// segments prefetch one batch ahead.  Ragged batches are padded, not predicated: a padding lane parks the pair
// (column TC, value 0) and column TC of the panel is an all-zero row, so every batch runs
// the branch-free 8-entry body and 0 * 0 never meets a user value.
template <int VEC>
__global__ __launch_bounds__(kCoreThreads, 4) void spmm_core_kernel(
    const int4 *__restrict__ work, const int32_t *__restrict__ tile_panel,
    const int64_t *__restrict__ tile_base, const int32_t *__restrict__ seg_off,
    const int32_t *__restrict__ ccol, const float *__restrict__ cval, const float *__restrict__ B,
    int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *panel = reinterpret_cast<float *>(smem);                                        // (TC+1) x 32 x VEC
    float2 *mpark = reinterpret_cast<float2 *>(smem + (size_t)(TC + 1) * 32 * VEC * 4);     // per wave 64 pairs

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & 31;
    const int gbase = lane & 32;
    const int group = threadIdx.x >> 5;
    const int4 wk = work[blockIdx.x];
    const int fcol0 = blockIdx.y * 32 * VEC;
    const int fcol = fcol0 + sub * VEC;
    const bool fact = fcol < f;
    float2 *mrow = mpark + wave * 64;
    const float2 *mg = mrow + gbase;

    float acc[RW][VEC];
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] = 0.f;

    if (threadIdx.x < 32) {   // the all-zero padding row
        float z[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] = 0.f;
        vstore<VEC>(panel + ((size_t)TC * 32 + threadIdx.x) * VEC, z);
    }

    for (int k = wk.y; k < wk.z; ++k) {
        const int64_t col0 = (int64_t)tile_panel[k] * TC;
        const int64_t base = tile_base[k];
        const int32_t *cb = ccol + base;   // uniform per workgroup
        const float *vb = cval + base;
        int32_t bound[RW + 1];
        {
            const int32_t *so = seg_off + (int64_t)k * (TR + 1) + group * RW;
#pragma unroll
            for (int j = 0; j <= RW; ++j) bound[j] = so[j];
        }
        // (1) first batch of every segment of this group
        int32_t pc[RW];
        float pv[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int len = bound[j + 1] - bound[j];
            const int e = (sub < len - 1) ? sub : len - 1;
            const uint32_t idx = (len > 0) ? (uint32_t)(bound[j] + e) : 0u;   // 32-bit lane offset, uniform base
            pc[j] = __builtin_nontemporal_load(cb + idx);
            pv[j] = __builtin_nontemporal_load(vb + idx);
        }
        __syncthreads();   // everyone is done reading the previous panel
        // stage B[col0 .. col0+TC) x [fcol0 .. fcol0+32*VEC) : one contiguous-row copy.  All
        // loads are issued before the first LDS write (clamped addresses, no branches: a
        // per-iteration bounds branch serialises the eight round trips); out-of-range
        // rows / features are zeroed by a select.
        {
            constexpr int NIT = TC * 32 / kCoreThreads;   // 8 row-vectors per thread
            constexpr int HB = PGCN_CORE_STAGE_BATCH;     // loads in flight per thread
            const int64_t lastrow = ncols - 1;
            const int lastf = f - VEC;
#pragma unroll
            for (int h = 0; h < NIT; h += HB) {
                float xs[HB][VEC];
#pragma unroll
                for (int it = 0; it < HB; ++it) {
                    const int idx = (h + it) * kCoreThreads + threadIdx.x;
                    const int r = idx >> 5, s = idx & 31;
                    const int64_t rr = (col0 + r < ncols) ? col0 + r : lastrow;
                    const int cc = (fcol0 + s * VEC < f) ? fcol0 + s * VEC : lastf;
                    vload<VEC>(xs[it], B + rr * ldb + cc);
                }
#pragma unroll
                for (int it = 0; it < HB; ++it) {
                    const int idx = (h + it) * kCoreThreads + threadIdx.x;
                    const int r = idx >> 5, s = idx & 31;
                    const bool ok = (col0 + r < ncols) && (fcol0 + s * VEC < f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) xs[it][v] = ok ? xs[it][v] : 0.f;
                    vstore<VEC>(panel + (size_t)idx * VEC, xs[it]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int len = bound[j + 1] - bound[j];
            int32_t cur_c = pc[j];
            float cur_v = pv[j];
            for (int b = 0; __any(b < len); b += 32) {
                const int cnt = len - b;   // entries left in this group's segment (may be <= 0)
                int32_t nx_c = 0;
                float nx_v = 0.f;
                if (__any(b + 32 < len)) {   // (2) one batch ahead
                    int e = b + 32 + sub;
                    e = (e < len - 1) ? e : len - 1;
                    const uint32_t idx = (len > 0) ? (uint32_t)(bound[j] + e) : 0u;
                    nx_c = __builtin_nontemporal_load(cb + idx);
                    nx_v = __builtin_nontemporal_load(vb + idx);
                }
                const bool valid = sub < cnt;
                __builtin_amdgcn_wave_barrier();
                mrow[lane] = make_float2(__int_as_float(valid ? cur_c : TC), valid ? cur_v : 0.f);
                __builtin_amdgcn_wave_barrier();
                // software-pipelined: the pairs of batch k4+CB are read while batch k4 computes, so a
                // batch costs one LDS round trip (rows) instead of two (pairs, then rows).
                float4 m[CB / 2];
#pragma unroll
                for (int u = 0; u < CB / 2; ++u) m[u] = *reinterpret_cast<const float4 *>(mg + 2 * u);
#pragma unroll
                for (int k4 = 0; k4 < 32; k4 += CB) {
                    if (!__any(k4 < cnt)) break;
                    float4 mn[CB / 2];
                    if (k4 + CB < 32) {
#pragma unroll
                        for (int u = 0; u < CB / 2; ++u) mn[u] = *reinterpret_cast<const float4 *>(mg + k4 + CB + 2 * u);
                    }
                    float x[CB][VEC];
#pragma unroll
                    for (int u = 0; u < CB / 2; ++u) {
                        lds_row<VEC>(x[2 * u], panel, __float_as_int(m[u].x), sub);
                        lds_row<VEC>(x[2 * u + 1], panel, __float_as_int(m[u].z), sub);
                    }
#pragma unroll
                    for (int u = 0; u < CB / 2; ++u) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[j][v] = fmaf(m[u].y, x[2 * u][v], acc[j][v]);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[j][v] = fmaf(m[u].w, x[2 * u + 1][v], acc[j][v]);
                    }
                    if (k4 + CB < 32) {
#pragma unroll
                        for (int u = 0; u < CB / 2; ++u) m[u] = mn[u];
                    }
                }
                cur_c = nx_c;
                cur_v = nx_v;
            }
        }
    }
    if (fact) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int rit = j * NG + group;   // row inside the tile
            vstore<VEC>(partial + ((int64_t)wk.w + rit) * f + fcol, acc[j]);
        }
    }
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
        static long pad = -1;   // experiment knob: extra dynamic LDS => one workgroup per CU
        if (pad < 0) { const char *e = getenv("PGCN_CORE_LDS_PAD"); pad = e ? atol(e) : 0; }
        smem += (size_t)pad;
        static bool attr_set = false;
        if (!attr_set) {
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_core_kernel<4>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_set = true;
        }
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
