// pgcn_spmm_dense3.hip -- dense 128 x 128 tiles of A on the bf16 matrix cores at fp32 accuracy.
//
// The fp32 MFMA (pgcn_spmm_dense.hip) runs at the fp32 VECTOR rate: v_mfma_f32_32x32x2_f32 is 64
// flop/clk/SIMD, 1/16 of v_mfma_f32_32x32x16_bf16.  Here every fp32 operand is written as the exact sum of
// THREE bf16 numbers -- x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2 (8 + 8 + 8
// significand bits; both remainders are exact fp32 subtractions and the last one is a bf16 number) -- and
// the product a.h is accumulated from the six partial products that matter,
//      a1 h1 + (a1 h2 + a2 h1) + (a1 h3 + a2 h2 + a3 h1),
// each of which is exact in fp32 (8 x 8 bits); the three dropped ones are below 2^-23 |a h|.  Six bf16 MFMAs
// replace sixteen-rate-units of fp32 MFMA: 2.7 x the fp32 matrix rate with the error class of an fp32 dot
// product (accumulation in fp32 inside the MFMA).  bf16 has the exponent range of fp32, so there is no
// scaling and no range cliff (an fp16 split would need both).
//
// Layout.  The tile's A planes are split and swizzled on the host into the A-operand order of the
// instruction (96 KB per tile):
//     planes[tile][w][ks][p][lane] (16 bytes) = bf16 plane p of A[32 w + (lane & 31)][16 ks + 8 (lane >> 5) + j], j = 0..7
// so wave w reads its operands of one k step as three coalesced 1 KB loads.  The feature panel is split on
// the fly: a thread reads 16 k values of ONE feature column (dword loads, coalesced across the lanes of a
// wave: no alignment or width requirement), splits them with v_cvt_pk_bf16_f32 and writes 8 consecutive k
// of a plane as one ds_write_b128 into a [feature][k] image -- the B-operand order, so a lane's operand
// is one ds_read_b128.  A quarter panel (32 k rows) is 3 x 128 x 80 B (rows padded by 16 B: the 16-lane
// groups of a ds_read_b128 and the 8-lane groups of a ds_write_b128 hit distinct banks); two quarter
// buffers = 60 KB, two workgroups per CU.  One barrier per quarter: the loads of quarter q + 1 are issued
// before the MFMAs of quarter q and split + written after them.
//
// Structural zeros: as in the fp32 kernel a non-finite sum sends the piece through an exact VALU path
// (products only where A != 0, operands straight from global memory).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pgcn_internal.h"

namespace {

constexpr int kT = 128;                  // tile edge
constexpr int kThreads = 256;
constexpr int kQ = 32;                   // k rows per staged quarter panel
constexpr int kRS = 2 * kQ + 16;         // bytes per feature row of a plane (64 B of bf16 + 16 B pad)
constexpr int kPL = kT * kRS;            // bytes per plane of a quarter
constexpr int kBuf = 3 * kPL;            // bytes per quarter buffer
constexpr size_t kSmem3 = 2 * (size_t)kBuf;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {     // {bf16(x) in bits 0-15, bf16(y) in bits 16-31}, RNE
    const f32x2 v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_as_f32(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_as_f32(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

__device__ __forceinline__ f32x16 mma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One pass over the tiles of a piece on the matrix cores.
template <int NBLK>
__device__ __forceinline__ void dense3_piece(const int4 wk, const int32_t *__restrict__ tile_panel,
                                             const u32x4 *__restrict__ planes, const float *__restrict__ B, int64_t ldb,
                                             int64_t ncols, int fcol0, int fw, char *smem, f32x16 (&acc)[NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    // staging role: feature column sn, k rows [8 sg, 8 sg + 16) of every quarter
    const int sn = threadIdx.x & (kT - 1);
    const int sg = (threadIdx.x >> 7) * 2;
    const bool sn_ok = sn < fw;
    const float *Bcol = B + fcol0 + (sn_ok ? sn : 0);
    float st[16];
    u32x4 a_cur[2][3], a_nxt[2][3];

    auto load_q = [&](int64_t ti, int q) {
        const int64_t r0 = (int64_t)tile_panel[ti] * kT + q * kQ + 8 * sg;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t r = r0 + i;
            st[i] = (sn_ok && r < ncols) ? Bcol[r * ldb] : 0.f;
        }
    };
    auto load_a = [&](int64_t ti, int q, u32x4 (&a)[2][3]) {
        const u32x4 *ap = planes + ((ti * 4 + w) * 8 + 2 * q) * 3 * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[s][p] = ap[(s * 3 + p) * 64];
    };
    auto store_q = [&](int buf) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 p1, p2, p3;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float x = st[8 * g + 2 * d], y = st[8 * g + 2 * d + 1];
                const uint32_t u1 = pack_bf16(x, y);
                const float rx = x - lo_as_f32(u1), ry = y - hi_as_f32(u1);       // exact
                const uint32_t u2 = pack_bf16(rx, ry);
                const uint32_t u3 = pack_bf16(rx - lo_as_f32(u2), ry - hi_as_f32(u2));   // exact, and a bf16 number
                p1[d] = u1; p2[d] = u2; p3[d] = u3;
            }
            char *dst = smem + buf * kBuf + sn * kRS + (sg + g) * 16;
            *reinterpret_cast<u32x4 *>(dst) = p1;
            *reinterpret_cast<u32x4 *>(dst + kPL) = p2;
            *reinterpret_cast<u32x4 *>(dst + 2 * kPL) = p3;
        }
    };
    auto compute_q = [&](int buf, const u32x4 (&a)[2][3]) {
        const char *base = smem + buf * kBuf + lo * kRS + hi * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 b[NBLK][3];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[nb][p] = *reinterpret_cast<const u32x4 *>(base + nb * 32 * kRS + s * 32 + p * kPL);
            // smallest terms first; the four column blocks interleaved (independent accumulators back to back)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mma(a[s][2], b[nb][0], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mma(a[s][0], b[nb][2], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mma(a[s][1], b[nb][1], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mma(a[s][1], b[nb][0], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mma(a[s][0], b[nb][1], acc[nb]);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[nb] = mma(a[s][0], b[nb][0], acc[nb]);
        }
    };

    const int nq = wk.z * 4;
    load_q(wk.y, 0);
    load_a(wk.y, 0, a_cur);
    store_q(0);
    for (int it = 0; it < nq; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nq;
        if (more) {
            load_q((int64_t)wk.y + ((it + 1) >> 2), (it + 1) & 3);
            load_a((int64_t)wk.y + ((it + 1) >> 2), (it + 1) & 3, a_nxt);
        }
        __syncthreads();       // buffer `buf` is complete; nobody reads buffer `buf ^ 1` (quarter it - 1) any more
        compute_q(buf, a_cur);
        if (more) {
            store_q(buf ^ 1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) a_cur[s][p] = a_nxt[s][p];
        }
    }
}

// Exact redo of a piece: products only where A != 0, k ascending, operands from global memory.
template <int NBLK>
__device__ __noinline__ void dense3_piece_exact(const int4 wk, const int32_t *__restrict__ tile_panel,
                                                const uint16_t *__restrict__ planes16, const float *__restrict__ B,
                                                int64_t ldb, int64_t ncols, int fcol0, int fw, f32x16 (&acc)[NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    for (int t = 0; t < wk.z; ++t) {
        const int64_t ti = (int64_t)wk.y + t;
        const int64_t prow0 = (int64_t)tile_panel[ti] * kT;
        for (int k = 0; k < kT; ++k) {
            const int ks = k >> 4, hk = (k >> 3) & 1, j = k & 7;
            float b[NBLK];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const int colj = nb * 32 + lo;
                b[nb] = (prow0 + k < ncols && colj < fw) ? B[(prow0 + k) * ldb + fcol0 + colj] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int64_t e = ((((ti * 4 + w) * 8 + ks) * 3) * 64 + hk * 32 + il) * 8 + j;
                const float a1 = __builtin_bit_cast(float, (uint32_t)planes16[e] << 16);
                const float a2 = __builtin_bit_cast(float, (uint32_t)planes16[e + 64 * 8] << 16);
                const float a3 = __builtin_bit_cast(float, (uint32_t)planes16[e + 2 * 64 * 8] << 16);
                const float x = (a1 + a2) + a3;      // exact: the planes are the split of one fp32 number
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) acc[nb][r] = x != 0.f ? fmaf(x, b[nb], acc[nb][r]) : acc[nb][r];
            }
        }
    }
}

// work: int4 {tile row, first tile, number of tiles, first slot}; NBLK = 32-column blocks holding features
template <int NBLK>
__global__ __launch_bounds__(kThreads, 2) void spmm_dense3_kernel(
    const int4 *__restrict__ work, const int32_t *__restrict__ tile_panel, const u32x4 *__restrict__ planes,
    const float *__restrict__ B, int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    const int4 wk = work[blockIdx.x];
    const int fcol0 = blockIdx.y * kT;
    const int fw = min(kT, f - fcol0);
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    dense3_piece<NBLK>(wk, tile_panel, planes, B, ldb, ncols, fcol0, fw, smem3, acc);
    bool bad = false;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) bad = bad || !(fabsf(acc[nb][r]) <= 3.402823466e+38f);
    if (__syncthreads_or(bad)) {
        f32x16 exact[NBLK];      // (its own array: the address of `acc` must not escape, or the accumulators live in scratch)
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) exact[nb][r] = 0.f;
        dense3_piece_exact<NBLK>(wk, tile_panel, reinterpret_cast<const uint16_t *>(planes), B, ldb, ncols, fcol0, fw, exact);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) acc[nb] = exact[nb];
    }
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

extern "C" int pgcn_spmm_dense_bf16x3_f32(const int32_t *work, int64_t nwork, const int32_t *tile_panel, const void *planes,
                                          const float *B, int64_t ldb, int64_t ncols, int32_t f, float *partial_ws,
                                          int64_t partial_ws_elems, int64_t nslots_total, pgcn_stream_t stream) {
    if (nwork < 0 || f <= 0 || ldb < f || ncols < 0) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: bad sizes");
    if (nwork == 0) return PGCN_OK;
    if (!work || !tile_panel || !planes || !B || !partial_ws)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: null pointer");
    if ((uintptr_t)planes % 16) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: planes must be 16-byte aligned");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_dense_bf16x3_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: work list too long");
    int dev = 0;
    PGCN_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set_dev[64] = {false};
    const bool attr_set = dev >= 0 && dev < 64 && attr_set_dev[dev];
    if (!attr_set) {
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
        if (dev >= 0 && dev < 64) attr_set_dev[dev] = true;
    }
    const int4 *w4 = reinterpret_cast<const int4 *>(work);
    const u32x4 *pl = reinterpret_cast<const u32x4 *>(planes);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nwork, (unsigned)((f + kT - 1) / kT)), block(kThreads);
    switch (((f < kT ? f : kT) + 31) / 32) {
        case 1: hipLaunchKernelGGL(spmm_dense3_kernel<1>, grid, block, kSmem3, s, w4, tile_panel, pl, B, ldb, ncols, f, partial_ws); break;
        case 2: hipLaunchKernelGGL(spmm_dense3_kernel<2>, grid, block, kSmem3, s, w4, tile_panel, pl, B, ldb, ncols, f, partial_ws); break;
        case 3: hipLaunchKernelGGL(spmm_dense3_kernel<3>, grid, block, kSmem3, s, w4, tile_panel, pl, B, ldb, ncols, f, partial_ws); break;
        default: hipLaunchKernelGGL(spmm_dense3_kernel<4>, grid, block, kSmem3, s, w4, tile_panel, pl, B, ldb, ncols, f, partial_ws); break;
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
