// pgcn_spmm_dense.hip -- the densest 128 x 128 tiles of A through the fp32 matrix cores.
//
// Degree-sorted power-law graphs have a corner of tiles that are 20-90 % full.  Through the
// LDS-gather kernel such a tile costs one 512 B LDS row read per stored entry (about 7.8 clk per
// entry and CU); as a dense 128 x 128 x f product on v_mfma_f32_32x32x2_f32 it costs 1024 MFMAs =
// 16.4 k clk per CU whatever its fill, so above ~15 % fill the matrix cores win.  fp32 in, fp32
// accumulate: the MFMA result is bit-for-bit a k-ordered fmaf chain (guide, "FP32-input MFMA"),
// so the numbers stay in the same error class as the other SpMM paths and the launch is
// deterministic.  This is not a reshaping of sparse work into a GEMM to reach the MFMA peak: only
// tiles that ARE dense take this path (PGCN_DENSE_TAU, default 0.20 of 16 384 entries).
//
// Layout.  A tile is stored dense and pre-swizzled into the MFMA A-operand order (64 KB):
//   vals[tile][w][s4][lane][e] = A[i = 32 w + (lane & 31)][k = 2 (4 s4 + e) + (lane >> 5)]
// so that wave w reads its operands as 16 fully coalesced float4 loads per tile (no LDS for A).
// The feature panel B[128 rows of the panel][f] is staged in LDS (64 KB) with the odd rows rotated
// by 32 columns: lanes 0-31 (row 2s) and 32-63 (row 2s+1) of a ds_read_b32 hit disjoint banks.
// A workgroup = 4 waves owns a piece = up to a few tiles of ONE tile row; wave w accumulates rows
// [32 w, 32 w + 32) x 128 features in 4 x 16 accumulator registers over all tiles of the piece and
// writes one 128 x f block of partial sums (combined by pgcn_spmm_fixup_f32, like core pieces).
//
// Panels are staged in halves with asynchronous global -> LDS copies (global_load_lds_dwordx4: no
// staging registers) and the tile loop is software-pipelined over the halves: the copies of the next
// half are in flight while the matrix cores work on the current one; two workgroups share a CU.
//
// Zeros of the dense tile are structural: 0 x Inf must not produce NaN where the sparse matrix has
// no entry.  A non-finite value in a panel makes every row of the tile's sums non-finite, so the
// accumulators are checked once per piece and a piece that fails is redone on an exact VALU path
// (products only where A != 0) -- slow, and never taken in a healthy training run.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pgcn_internal.h"
#include "pgcn_once.h"

namespace {

constexpr int kT = 128;                 // tile edge (PGCN_CORE_TR / PGCN_CORE_TC)
constexpr int kThreads = 256;
constexpr size_t kSmem = (size_t)kT * kT * sizeof(float);

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Stage rows [64 h, 64 h + 64) of a tile's feature panel into their half of the LDS image.
__device__ __forceinline__ void stage_half(const float *__restrict__ B, int64_t ldb, int64_t ncols, int64_t prow0,
                                           int fcol0, int fw, int vec, int h, int w, float *smem) {
    float4 *s4p = reinterpret_cast<float4 *>(smem) + h * (kT / 2) * 32;
    const int64_t r0 = prow0 + h * (kT / 2);
    if (vec && fw == kT && r0 + kT / 2 <= ncols) {
        // full half panel: 8 asynchronous global -> LDS copies per thread, no staging registers.  The LDS
        // image is lane-linear, so the rotation of the odd rows is applied to the SOURCE column.
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = q * kThreads + (int)threadIdx.x;
            const int row = idx >> 5, c4 = ((idx & 31) - 8 * (row & 1)) & 31;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(B + (r0 + row) * ldb + fcol0 + c4 * 4),
                (__attribute__((address_space(3))) void *)(s4p + q * kThreads + w * 64), 16, 0, 0);
        }
    } else if (vec) {
        const int f4 = fw >> 2;
#pragma unroll 4
        for (int q = 0; q < 8; ++q) {
            const int idx = q * kThreads + (int)threadIdx.x;
            const int row = idx >> 5, c4 = idx & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < ncols && c4 < f4) v = *reinterpret_cast<const float4 *>(B + (r0 + row) * ldb + fcol0 + c4 * 4);
            s4p[row * 32 + ((c4 + 8 * (row & 1)) & 31)] = v;
        }
    } else {                             // any width / alignment: scalar staging
        float *sp = smem + h * (kT / 2) * kT;
        for (int idx = threadIdx.x; idx < (kT / 2) * kT; idx += kThreads) {
            const int row = idx >> 7, c = idx & 127;
            float v = 0.f;
            if (r0 + row < ncols && c < fw) v = B[(r0 + row) * ldb + fcol0 + c];
            sp[row * kT + ((c + 32 * (row & 1)) & 127)] = v;
        }
    }
}

// k rows [64 h, 64 h + 64) of one tile.  EXACT = false: matrix cores, `a` = this wave's 8 float4 of
// A operands of the half.  EXACT = true: products only where A != 0 (k ascending, fmaf).
template <int NBLK, bool EXACT>
__device__ __forceinline__ void compute_half(const float4 (&a)[8], const float *__restrict__ tv, int h, int w, int hi,
                                             int lo, const float *smem, f32x16 (&acc)[NBLK]) {
    if (!EXACT) {
        // (reading the B operands of step t + 1 before the MFMAs of step t -- sched_barrier-pinned -- was measured
        //  in r03: no change, the compiler's order stays)
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const float av4[4] = {a[s4].x, a[s4].y, a[s4].z, a[s4].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 64 * h + 2 * (4 * s4 + e) + hi;
                const float *brow = smem + k * kT;
                float b[NBLK];
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) b[nb] = brow[(nb * 32 + lo + 32 * hi) & 127];
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av4[e], b[nb], acc[nb], 0, 0, 0);
            }
        }
    } else {
        // this lane owns D[row = (r & 3) + 8 (r >> 2) + 4 hi][col = lo] of every 32 x 32 block
        for (int k = 64 * h; k < 64 * h + 64; ++k) {
            const int s = k >> 1, kh = k & 1;
            float b[NBLK];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) b[nb] = smem[k * kT + ((nb * 32 + lo + 32 * kh) & 127)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float x = tv[((w * 16 + (s >> 2)) * 64 + kh * 32 + il) * 4 + (s & 3)];
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) acc[nb][r] = x != 0.f ? fmaf(x, b[nb], acc[nb][r]) : acc[nb][r];
            }
        }
    }
}

// One pass over the tiles of a piece, software-pipelined over half panels: while the matrix cores
// work on one half of the LDS image, the asynchronous copies of the next half (of this tile or of
// the next one) and the loads of its A operands are in flight; one barrier per half.
template <int NBLK, bool EXACT>
__device__ __forceinline__ void dense_piece(const int4 wk, const int32_t *__restrict__ tile_panel,
                                            const float *__restrict__ vals, const float *__restrict__ B, int64_t ldb,
                                            int64_t ncols, int fcol0, int fw, int vec, float *smem,
                                            f32x16 (&acc)[NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    float4 a0[8], a1[8];
    auto load_a = [&](int64_t ti, int h, float4 (&a)[8]) {
        if (!EXACT) {
            const float4 *av = reinterpret_cast<const float4 *>(vals + ti * (int64_t)(kT * kT)) + (w * 16 + 8 * h) * 64 + lane;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) a[s4] = av[s4 * 64];
        }
    };
    __syncthreads();                     // (second pass: the LDS image of the first pass is no longer read)
    stage_half(B, ldb, ncols, (int64_t)tile_panel[wk.y] * kT, fcol0, fw, vec, 0, w, smem);
    load_a(wk.y, 0, a0);
    for (int t = 0; t < wk.z; ++t) {
        const int64_t ti = (int64_t)wk.y + t;
        const int64_t prow0 = (int64_t)tile_panel[ti] * kT;
        const float *tv = vals + ti * (int64_t)(kT * kT);
        __syncthreads();                 // half 0 has landed; nobody reads half 1 of the previous tile any more
        stage_half(B, ldb, ncols, prow0, fcol0, fw, vec, 1, w, smem);
        load_a(ti, 1, a1);
        compute_half<NBLK, EXACT>(a0, tv, 0, w, hi, lo, smem, acc);
        __syncthreads();                 // half 1 has landed; nobody reads half 0 any more
        if (t + 1 < wk.z) {
            stage_half(B, ldb, ncols, (int64_t)tile_panel[ti + 1] * kT, fcol0, fw, vec, 0, w, smem);
            load_a(ti + 1, 0, a0);
        }
        compute_half<NBLK, EXACT>(a1, tv, 1, w, hi, lo, smem, acc);
    }
}

// work: int4 {tile row, first tile, number of tiles, first slot}; NBLK = 32-column blocks holding features
template <int NBLK>
__global__ __launch_bounds__(kThreads, 2) void spmm_dense_kernel(
    const int4 *__restrict__ work, const int32_t *__restrict__ tile_panel, const float *__restrict__ vals,
    const float *__restrict__ B, int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial, int vec) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int4 wk = work[blockIdx.x];
    const int fcol0 = blockIdx.y * kT;   // this workgroup's 128 feature columns
    const int fw = min(kT, f - fcol0);
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    dense_piece<NBLK, false>(wk, tile_panel, vals, B, ldb, ncols, fcol0, fw, vec, smem, acc);
    // 0 x Inf of a structural zero shows up as a non-finite sum: redo the piece exactly (a finite
    // overflow lands here too and simply comes out the same, slower)
    bool bad = false;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) bad = bad || !(fabsf(acc[nb][r]) <= 3.402823466e+38f);
    if (__syncthreads_or(bad)) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        dense_piece<NBLK, true>(wk, tile_panel, vals, B, ldb, ncols, fcol0, fw, vec, smem, acc);
    }
    // partial sums: slot row = first slot + 32 w + D row
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) {
        const int colj = nb * 32 + lo;
        if (colj < fw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                partial[((int64_t)wk.w + 32 * w + il) * f + fcol0 + colj] = acc[nb][r];
            }
        }
    }
}

}  // namespace

extern "C" int pgcn_spmm_dense_f32(const int32_t *work, int64_t nwork, const int32_t *tile_panel, const float *vals,
                                   const float *B, int64_t ldb, int64_t ncols, int32_t f, float *partial_ws,
                                   int64_t partial_ws_elems, int64_t nslots_total, pgcn_stream_t stream) {
    if (nwork < 0 || f <= 0 || ldb < f || ncols < 0) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_f32: bad sizes");
    if (nwork == 0) return PGCN_OK;
    if ((uintptr_t)vals % 16) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_f32: vals must be 16-byte aligned");
    const int vec = (f % 4 == 0 && (uintptr_t)B % 16 == 0 && ldb % 4 == 0) ? 1 : 0;
    if (!work || !tile_panel || !vals || !B || !partial_ws)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_f32: null pointer");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_dense_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_f32: work list too long");
    static PgcnPerDeviceOnce once;
    if (int rc = once.run([&]() -> int {
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
            return PGCN_OK;
        }))
        return rc;
    const int4 *w4 = reinterpret_cast<const int4 *>(work);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nwork, (unsigned)((f + kT - 1) / kT)), block(kThreads);
    switch (((f < kT ? f : kT) + 31) / 32) {
        case 1: hipLaunchKernelGGL(spmm_dense_kernel<1>, grid, block, kSmem, s, w4, tile_panel, vals, B, ldb, ncols, f, partial_ws, vec); break;
        case 2: hipLaunchKernelGGL(spmm_dense_kernel<2>, grid, block, kSmem, s, w4, tile_panel, vals, B, ldb, ncols, f, partial_ws, vec); break;
        case 3: hipLaunchKernelGGL(spmm_dense_kernel<3>, grid, block, kSmem, s, w4, tile_panel, vals, B, ldb, ncols, f, partial_ws, vec); break;
        default: hipLaunchKernelGGL(spmm_dense_kernel<4>, grid, block, kSmem, s, w4, tile_panel, vals, B, ldb, ncols, f, partial_ws, vec); break;
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
