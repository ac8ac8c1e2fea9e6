// pgcn_spmm_dense3.hip -- dense 128 x 128 tiles of A on the bf16 matrix cores at fp32 accuracy.
// EXPERIMENT (r03, tools/experiments/dense3): measured on the stand-alone harness only -- 7.75 us per tile and CU
// against 10.98 for the fp32-MFMA kernel of the library, same error class, all edge cases green
// (profiles/r03_dense3_bench.txt); not in libpgcn_hip.so until it is measured inside the launch group
// (integrate.patch wires it in behind tuning.dense_bf16x3).
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

#if __has_include("pgcn_internal.h")       // (copied into csrc/ by the integration)
#include "pgcn_internal.h"
#else
#include "../../../scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd/csrc/pgcn_internal.h"
#endif

namespace {

constexpr int kT = 128;                  // tile edge
constexpr int kThreads = 256;
constexpr int kQ = 32;                   // k rows per staged quarter panel
constexpr int kRS = 2 * kQ + 16;         // bytes per feature row of a plane (64 B of bf16 + 16 B pad)
constexpr int kPL = kT * kRS;            // bytes per plane of a quarter
constexpr int kBuf = 3 * kPL;            // bytes per quarter buffer
constexpr size_t kSmem3 = 2 * (size_t)kBuf;
// tools/micro/dense3_bench.cpp compiles this file a second and third time with PGCN_DENSE3_PROBE = 1 (the panel is
// staged once per piece: no loads / splits / LDS writes in the loop) and 2 (no A loads either): timing only, wrong
// sums -- where the time of a tile goes.  The library is built with 0: the branches below fold away.
#ifndef PGCN_DENSE3_PROBE
#define PGCN_DENSE3_PROBE 0
#endif
constexpr int kProbe = PGCN_DENSE3_PROBE;
// PGCN_DENSE3_PROBE = 3: the real kernel with per-wave phase timers (s_memtime ticks = shader cycles): lane 0 of every
// wave adds up, over the quarters of its piece, {requests issued, barrier wait, MFMA block, split + LDS writes} and
// stores them with the prologue and the whole loop to timers[piece][wave][6] (pgcn_dense3_set_timers).
#if PGCN_DENSE3_PROBE == 3
__device__ unsigned long long *g_dense3_timers = nullptr;
#endif
__device__ __forceinline__ unsigned long long tick() {
#if PGCN_DENSE3_PROBE == 3
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    return t;
#else
    return 0;
#endif
}

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

// x, y -> the three bf16 planes of both, packed {x in bits 0-15, y in bits 16-31}
__device__ __forceinline__ void split_pair(float x, float y, uint32_t &u1, uint32_t &u2, uint32_t &u3) {
    u1 = pack_bf16(x, y);
    const float rx = x - lo_as_f32(u1), ry = y - hi_as_f32(u1);          // exact
    u2 = pack_bf16(rx, ry);
    u3 = pack_bf16(rx - lo_as_f32(u2), ry - hi_as_f32(u2));              // exact, and a bf16 number
}

// One pass over the tiles of a piece on the matrix cores.  The loop runs over QUARTER panels (32 k rows), two per
// trip so that registers alternate without copies:
//   step q:  request the panel rows of quarter q + 2 (consumed at the end of step q + 1: two steps of cover) and the
//            A operands of quarter q + 1 | barrier | MFMAs of quarter q from LDS buffer q & 1 | split the rows of
//            quarter q + 1 and write them to buffer (q + 1) & 1.
template <int NBLK>
__device__ __forceinline__ void dense3_piece(const int4 wk, const int32_t *__restrict__ tile_panel,
                                             const u32x4 *__restrict__ planes, const float *__restrict__ B, int64_t ldb,
                                             int64_t ncols, int fcol0, int fw, char *smem, f32x16 (&acc)[NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    // staging role: feature column sn, k rows [8 sg, 8 sg + 16) of every quarter (sg is the same for a whole wave)
    const int sn = threadIdx.x & (kT - 1);
    const int sg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)) * 2;
    const bool sn_ok = sn < fw;
    const uint32_t coff = (uint32_t)(fcol0 + (sn_ok ? sn : 0));
    float st0[16], st1[16];
    u32x4 aA[2][3], aB[2][3];            // bf16 planes [k step][plane]

    auto row0 = [&](int qi) -> int64_t {              // first operand row of this wave's share of quarter qi (wave-uniform)
        return (int64_t)tile_panel[(int64_t)wk.y + (qi >> 2)] * kT + (qi & 3) * kQ + 8 * sg;
    };
    auto load_q = [&](int qi, float (&st)[16]) {      // unconditional loads from clamped rows: masked when stored
        const int64_t r0 = row0(qi);
        if (r0 + 16 <= ncols) {                       // (wave-uniform branch) all rows exist: one scalar add per row
            const float *rowp = B + r0 * ldb;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                st[i] = rowp[coff];
                rowp += ldb;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int64_t r = r0 + i;
                r = r < ncols ? r : ncols - 1;
                st[i] = (B + r * ldb)[coff];
            }
        }
    };
    auto load_a = [&](int qi, u32x4 (&a)[2][3]) {
        const int64_t ti = (int64_t)wk.y + (qi >> 2);
        const u32x4 *ap = planes + ((ti * 4 + w) * 8 + 2 * (qi & 3)) * 3 * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[s][p] = ap[(s * 3 + p) * 64];
    };
    auto store_q = [&](int qi, const float (&st)[16], int buf) {
        const int64_t left = ncols - row0(qi);                       // rows of this share that exist (wave-uniform)
        const bool full = left >= 16 && fw == kT;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            u32x4 p1, p2, p3;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float x = st[8 * g + 2 * d], y = st[8 * g + 2 * d + 1];
                if (!full) {
                    x = (sn_ok && 8 * g + 2 * d < left) ? x : 0.f;
                    y = (sn_ok && 8 * g + 2 * d + 1 < left) ? y : 0.f;
                }
                uint32_t u1, u2, u3;
                split_pair(x, y, u1, u2, u3);
                p1[d] = u1; p2[d] = u2; p3[d] = u3;
            }
            char *dst = smem + buf * kBuf + sn * kRS + (sg + g) * 16;
            *reinterpret_cast<u32x4 *>(dst) = p1;
            *reinterpret_cast<u32x4 *>(dst + kPL) = p2;
            *reinterpret_cast<u32x4 *>(dst + 2 * kPL) = p3;
        }
    };
    // MFMAs of one quarter.  The B operands are read HALF A K STEP ahead (two column blocks x three planes = six
    // ds_read_b128 in flight under twelve MFMAs): the scheduler left to itself keeps one operand in flight and
    // waits for the LDS before almost every MFMA (r03 harness: 41 % of the matrix rate with nothing else in the loop).
    auto compute_q = [&](int buf, const u32x4 (&a)[2][3]) {
        const char *base = smem + buf * kBuf + lo * kRS + hi * 16;
        constexpr int NH = (NBLK + 1) / 2;            // half steps per k step: pairs of 32-column blocks
        u32x4 b[2][2][3];
        auto rd = [&](int t, u32x4 (&bb)[2][3]) {
            const int s = t / NH, h = t % NH;
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (2 * h + e < NBLK) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        bb[e][p] = *reinterpret_cast<const u32x4 *>(base + (2 * h + e) * 32 * kRS + s * 32 + p * kPL);
                }
        };
        auto mm = [&](int t, const u32x4 (&bb)[2][3]) {
            const int s = t / NH, h = t % NH;
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};     // smallest terms first
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    if (2 * h + e < NBLK) acc[2 * h + e] = mma(a[s][pa[i]], bb[e][pb[i]], acc[2 * h + e]);
        };
        rd(0, b[0]);
#pragma unroll
        for (int t = 0; t < 2 * NH; ++t) {
            if (t + 1 < 2 * NH) rd(t + 1, b[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0x16);     // VALU / SALU / VMEM may move across, LDS reads and MFMAs may not
            mm(t, b[t & 1]);
            __builtin_amdgcn_sched_barrier(0x16);
        }
    };

    const int nq = wk.z * 4;                          // even, >= 4
    unsigned long long tm[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tp = tick();
    load_q(0, st0);
    load_a(0, aA);
    load_q(1, st1);
    store_q(0, st0, 0);
    unsigned long long t0 = tick();
    tm[4] = t0 - tp;
    for (int it = 0; it < nq; it += 2) {
        // quarter it: buffer 0, operands aA
        if (kProbe != 1 && kProbe != 2 && it + 2 < nq) load_q(it + 2, st0);
        if (kProbe != 2) load_a(it + 1, aB);
        unsigned long long t1 = tick();
        __syncthreads();       // buffer 0 is complete; nobody reads buffer 1 (quarter it - 1) any more
        unsigned long long t2 = tick();
        compute_q(0, aA);
        unsigned long long t3 = tick();
        if (kProbe != 1 && kProbe != 2) store_q(it + 1, st1, 1);
        unsigned long long t4 = tick();
        tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += t3 - t2; tm[3] += t4 - t3;
        // quarter it + 1: buffer 1, operands aB
        if (kProbe != 1 && kProbe != 2 && it + 3 < nq) load_q(it + 3, st1);
        if (kProbe != 2 && it + 2 < nq) load_a(it + 2, aA);
        t1 = tick();
        __syncthreads();
        t2 = tick();
        compute_q((kProbe == 1 || kProbe == 2) ? 0 : 1, kProbe != 2 ? aB : aA);
        t3 = tick();
        if (kProbe != 1 && kProbe != 2 && it + 2 < nq) store_q(it + 2, st0, 0);
        t0 = tick();
        tm[0] += t1 - t4; tm[1] += t2 - t1; tm[2] += t3 - t2; tm[3] += t0 - t3;
    }
#if PGCN_DENSE3_PROBE == 3
    tm[5] = t0 - tp;
    if (lane == 0 && g_dense3_timers) {
        unsigned long long *o = g_dense3_timers + ((size_t)blockIdx.x * 4 + w) * 6;
        for (int i = 0; i < 6; ++i) o[i] = tm[i];
    }
#endif
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

#if PGCN_DENSE3_PROBE == 3
extern "C" int pgcn_dense3_set_timers(void *buf) {           // buf: npieces x 4 waves x 6 uint64 (device memory), or null
    unsigned long long *p = static_cast<unsigned long long *>(buf);
    PGCN_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_dense3_timers), &p, sizeof(p)));
    return PGCN_OK;
}
#endif
