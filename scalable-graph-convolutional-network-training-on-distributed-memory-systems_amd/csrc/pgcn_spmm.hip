// pgcn_spmm.hip -- CSR SpMM for gfx950 (MI355X / CDNA4), fp32.
//
//   C[nrows x f] (+)= A[nrows x *] . B[* x f]
//
// Replaces torch.sparse.mm at /root/reference/GPU/PGCN.py:127,132 (== GrB_mxm on
// PLUS_TIMES_FP32, Parallel-GCN/main.c:271,295,376,400).
//
// Mapping to the machine (wave = 64 lanes, no tensor-core reshaping: this is
// HBM / cache-bandwidth bound gather work):
//   * a GROUP of LPR lanes owns one task (= one row, or one <=chunk segment of
//     a long row); each lane of the group owns VEC consecutive features, so a
//     group reads one dense row of B as one fully coalesced LPR*VEC*4-byte
//     segment (f=128: 32 lanes x 16 B = 512 B; two tasks per wave make the
//     1 KiB global_load_dwordx4 the L1 likes).  G = 64/LPR tasks per wave.
//   * (col,val) pairs of a task are fetched LPR at a time, one pair per lane
//     (coalesced, non-temporal: they are streamed exactly once and must not
//     evict feature rows from the per-XCD L2), then broadcast inside the
//     group with ds_bpermute (LDS crossbar, no LDS memory).
//   * the gather loop is unrolled x8 so each wave keeps eight independent
//     row-loads in flight; fp32 FMA accumulation in registers; no atomics.
//   * rows longer than the plan's chunk are split; segment partial sums go to
//     a caller-provided work-space and are combined in fixed order by
//     spmm_fixup_kernel (deterministic).
//   * XCD swizzle: block b runs on XCD b%8 (private 4 MiB L2); the optional
//     remap hands each XCD one contiguous range of rows so neighbouring rows
//     (which share columns in any locality-preserving ordering) share an L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgcn_internal.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * 64;
constexpr int kUnroll = 8;

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void vload(float (&x)[VEC], const float *p) {
    using V = typename VecT<VEC>::type;
    const V v = *reinterpret_cast<const V *>(p);
    if constexpr (VEC == 1) { x[0] = v; }
    if constexpr (VEC == 2) { x[0] = v.x; x[1] = v.y; }
    if constexpr (VEC == 4) { x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
}

template <int VEC>
__device__ __forceinline__ void vstore(float *p, const float (&x)[VEC]) {
    using V = typename VecT<VEC>::type;
    if constexpr (VEC == 1) { *p = x[0]; }
    if constexpr (VEC == 2) { *reinterpret_cast<V *>(p) = make_float2(x[0], x[1]); }
    if constexpr (VEC == 4) { *reinterpret_cast<V *>(p) = make_float4(x[0], x[1], x[2], x[3]); }
}

__device__ __forceinline__ int64_t swizzle_block(int64_t b, int64_t nb, bool on) {
    if (!on) return b;
    // block b lands on XCD b % 8; give XCD x the contiguous block range
    // [x*per, (x+1)*per).  Bijective for any nb via the ceil-sized ranges +
    // bounds check in the caller (ids >= nb simply have no task).
    const int64_t per = (nb + 7) / 8;
    return (b % 8) * per + b / 8;
}

// tasks: int4 {row, offset within row, length, slot (-1 = write C directly)}
template <int LPR, int VEC>
__global__ __launch_bounds__(kThreads) void spmm_tasks_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int4 *__restrict__ tasks, int64_t ntasks,
    const int32_t *__restrict__ row_map, const float *__restrict__ B, int64_t ldb,
    float *__restrict__ C, int64_t ldc, int32_t f, float *__restrict__ partial,
    int64_t nblocks, uint32_t flags) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane / LPR;
    const int sub = lane % LPR;
    const int gbase = grp * LPR;

    const int64_t bid = swizzle_block(blockIdx.x, nblocks, (flags & PGCN_SPMM_XCD_SWIZZLE) != 0);
    const int64_t tid = (bid * kWavesPerBlock + wave) * G + grp;
    const int fcol = (blockIdx.y * LPR + sub) * VEC;  // first feature owned by this lane
    const bool fact = fcol < f;

    int32_t row = 0, len = 0, slot = -1;
    int64_t kbeg = 0;
    if (tid < ntasks) {
        if (tasks) {
            const int4 t = tasks[tid];
            row = t.x; len = t.z; slot = t.w;
            kbeg = rowptr[row] + t.y;
        } else {  // one task per row
            row = (int32_t)tid;
            kbeg = rowptr[row];
            len = (int32_t)(rowptr[row + 1] - kbeg);
        }
    }
    const bool tact = tid < ntasks;

    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

    const float *Bl = B + fcol;

    for (int base = 0; __any(base < len); base += LPR) {
        // one (col,val) pair per lane, streamed once -> non-temporal
        int32_t my_c = 0;
        float my_v = 0.f;
        const int e = base + sub;
        if (e < len) {
            my_c = __builtin_nontemporal_load(col + kbeg + e);
            my_v = val ? __builtin_nontemporal_load(val + kbeg + e) : 1.f;
        }
        const int cnt = len - base;  // edges left for this group (may be <= 0)
        for (int k = 0; k < LPR; k += kUnroll) {
            if (!__any(k < cnt)) break;
            float x[kUnroll][VEC];
            float w[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (k + u < LPR) {  // compile-time for LPR < kUnroll
                    const int src = gbase + k + u;
                    const int32_t c = __shfl(my_c, src);
                    w[u] = __shfl(my_v, src);
                    const bool p = fact && (k + u < cnt);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) x[u][v] = 0.f;
                    if (p) vload<VEC>(x[u], Bl + (int64_t)c * ldb);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (k + u < LPR) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w[u], x[u][v], acc[v]);
                }
            }
        }
    }

    if (tact && fact) {
        if (slot >= 0) {
            vstore<VEC>(partial + (int64_t)slot * f + fcol, acc);
        } else {
            const int64_t orow = row_map ? row_map[row] : row;
            float *c = C + orow * ldc + fcol;
            if (flags & PGCN_SPMM_ACCUMULATE) {
                float old[VEC];
                vload<VEC>(old, c);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += old[v];
            }
            vstore<VEC>(c, acc);
        }
    }
}

// fix: int4 {row, first slot, #segments, unused}; sums the segments of a split
// row in slot order -> deterministic.
template <int LPR, int VEC>
__global__ __launch_bounds__(kThreads) void spmm_fixup_kernel(
    const int4 *__restrict__ fix, int64_t nfix, const int32_t *__restrict__ row_map,
    const float *__restrict__ partial, float *__restrict__ C, int64_t ldc, int32_t f,
    uint32_t flags) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane / LPR;
    const int sub = lane % LPR;
    const int64_t id = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * G + grp;
    const int fcol = (blockIdx.y * LPR + sub) * VEC;
    if (id >= nfix || fcol >= f) return;
    const int4 t = fix[id];
    const int64_t orow = row_map ? row_map[t.x] : t.x;
    float *c = C + orow * ldc + fcol;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (flags & PGCN_SPMM_ACCUMULATE) vload<VEC>(acc, c);
    for (int s = 0; s < t.z; ++s) {
        float x[VEC];
        vload<VEC>(x, partial + (int64_t)(t.y + s) * f + fcol);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += x[v];
    }
    vstore<VEC>(c, acc);
}

struct Shape { int lpr, vec; };

Shape pick_shape(int32_t f, const void *B, int64_t ldb, const void *C, int64_t ldc,
                 const void *ws) {
    const bool al16 = ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) &&
                      ((uintptr_t)ws % 16 == 0) && ldb % 4 == 0 && ldc % 4 == 0 && f % 4 == 0;
    const int vec = al16 ? 4 : 1;
    const int nv = (f + vec - 1) / vec;
    int lpr = 1;
    while (lpr < nv && lpr < 64) lpr *= 2;
    return {lpr, vec};
}

template <int LPR, int VEC>
int launch(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *tasks,
           int64_t ntasks, const int32_t *fix, int64_t nfix, const int32_t *row_map,
           const float *B, int64_t ldb, float *C, int64_t ldc, int32_t f, float *partial,
           uint32_t flags, hipStream_t s) {
    constexpr int G = 64 / LPR;
    const int ntiles = (f + LPR * VEC - 1) / (LPR * VEC);
    if (ntasks > 0) {
        const int64_t per_block = (int64_t)kWavesPerBlock * G;
        int64_t nblocks = (ntasks + per_block - 1) / per_block;
        int64_t grid = nblocks;
        if (flags & PGCN_SPMM_XCD_SWIZZLE) grid = ((nblocks + 7) / 8) * 8;
        if (grid > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "spmm: too many tasks for one launch");
        hipLaunchKernelGGL((spmm_tasks_kernel<LPR, VEC>), dim3((unsigned)grid, ntiles), dim3(kThreads),
                           0, s, rowptr, col, val, reinterpret_cast<const int4 *>(tasks), ntasks,
                           row_map, B, ldb, C, ldc, f, partial, grid, flags);
        PGCN_HIP_CHECK(hipGetLastError());
    }
    if (nfix > 0) {
        const int64_t per_block = (int64_t)kWavesPerBlock * G;
        const int64_t grid = (nfix + per_block - 1) / per_block;
        hipLaunchKernelGGL((spmm_fixup_kernel<LPR, VEC>), dim3((unsigned)grid, ntiles), dim3(kThreads),
                           0, s, reinterpret_cast<const int4 *>(fix), nfix, row_map, partial, C, ldc,
                           f, flags);
        PGCN_HIP_CHECK(hipGetLastError());
    }
    return PGCN_OK;
}

int dispatch(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *tasks,
             int64_t ntasks, const int32_t *fix, int64_t nfix, const int32_t *row_map,
             const float *B, int64_t ldb, float *C, int64_t ldc, int32_t f, float *partial,
             uint32_t flags, hipStream_t s) {
    const Shape sh = pick_shape(f, B, ldb, C, ldc, partial);
#define PGCN_CASE(L, V)                                                                       \
    if (sh.lpr == L && sh.vec == V)                                                           \
        return launch<L, V>(rowptr, col, val, tasks, ntasks, fix, nfix, row_map, B, ldb, C,   \
                            ldc, f, partial, flags, s);
    PGCN_CASE(1, 4) PGCN_CASE(2, 4) PGCN_CASE(4, 4) PGCN_CASE(8, 4) PGCN_CASE(16, 4)
    PGCN_CASE(32, 4) PGCN_CASE(64, 4)
    PGCN_CASE(1, 1) PGCN_CASE(2, 1) PGCN_CASE(4, 1) PGCN_CASE(8, 1) PGCN_CASE(16, 1)
    PGCN_CASE(32, 1) PGCN_CASE(64, 1)
#undef PGCN_CASE
    return pgcn_set_error(PGCN_EINVAL, "spmm: no kernel shape");
}

}  // namespace

extern "C" int pgcn_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                 int64_t nrows, const float *B, int64_t ldb, float *C,
                                 int64_t ldc, int32_t f, uint32_t flags, pgcn_stream_t stream) {
    if (nrows < 0 || f <= 0 || ldb < f || ldc < f)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_f32: bad sizes");
    if (nrows == 0) return PGCN_OK;
    if (!rowptr || !col || !B || !C)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_f32: null pointer");
    if (nrows > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_f32: nrows >= 2^31");
    return dispatch(rowptr, col, val, nullptr, nrows, nullptr, 0, nullptr, B, ldb, C, ldc, f,
                    nullptr, flags, (hipStream_t)stream);
}

extern "C" int pgcn_spmm_csr_plan_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                      const int32_t *tasks, int64_t ntasks, const int32_t *fix,
                                      int64_t nfix, const int32_t *row_map, const float *B,
                                      int64_t ldb, float *C, int64_t ldc, int32_t f,
                                      float *partial_ws, int64_t partial_ws_elems, int64_t nslots,
                                      uint32_t flags, pgcn_stream_t stream) {
    if (ntasks < 0 || nfix < 0 || f <= 0 || ldb < f || ldc < f)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_plan_f32: bad sizes");
    if (ntasks == 0) return PGCN_OK;
    if (!rowptr || !col || !B || !C || !tasks || (nfix > 0 && (!fix || !partial_ws)))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_plan_f32: null pointer");
    if (nslots < 0 || (nfix > 0 && partial_ws_elems < nslots * (int64_t)f))
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_csr_plan_f32: partial work-space too small");
    return dispatch(rowptr, col, val, tasks, ntasks, fix, nfix, row_map, B, ldb, C, ldc, f,
                    partial_ws, flags, (hipStream_t)stream);
}
