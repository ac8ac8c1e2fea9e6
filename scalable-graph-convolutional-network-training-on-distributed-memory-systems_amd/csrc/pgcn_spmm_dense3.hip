// pgcn_spmm_dense3.hip -- the dense 128 x 128 tiles of A on the bf16 matrix cores at fp32 accuracy (r04).
//
// The fp32 MFMA of pgcn_spmm_dense.hip (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate, 1/16 of
// v_mfma_f32_32x32x16_bf16.  Here every fp32 operand is the exact sum of THREE bf16 numbers,
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2      (8 + 8 + 8 significand bits;
// both remainders are exact fp32 subtractions and the last one is a bf16 number), and a product a.h is accumulated
// from the six partial products that matter, smallest first,
//     a3 h1 + a1 h3 + a2 h2 + a2 h1 + a1 h2 + a1 h1,
// each exact in fp32 (8 x 8 bits), accumulated in fp32 inside the MFMA; the three dropped ones are below
// 2^-23 |a h|.  Error class of an fp32 dot product (harness: 2.9e-7 of sum |a||h| against 2.4e-7 of the fp32 MFMA
// path), bf16 has the exponent range of fp32 (no scaling, no range cliff), deterministic.  Six bf16 MFMAs replace
// sixteen rate units of fp32 MFMA (replaces part of /root/reference/GPU/PGCN.py:127 torch.sparse.mm).
//
// What the r03 harness version taught (tools/experiments/dense3, 7.9 us per tile and CU against 10.9 for fp32):
// splitting the feature panel INSIDE the tile loop costs more than the MFMAs (per quarter panel and wave: 2 075
// ticks to issue the dword loads + 1 070 to split and write LDS against 1 536 matrix-pipe cycles), and it is redone
// by every tile that shares the panel (~16 x on the benchmark graph).  So:
//   * split_panels_kernel splits the panels that dense tiles refer to ONCE per SpMM (490 of 1 821 panels on the
//     benchmark graph: 31 MB read, 47 MB written) into the exact LDS image of the tile kernel, in a work-space:
//         image[panel][feature block] = [quarter q (32 k)][plane p][k group kg (8 k)][column n (128)] x 16 B
//     (8 bf16: k = 32 q + 8 kg + j), i.e. a lane's B operand of v_mfma_f32_32x32x16_bf16 is one 16-byte slot and
//     the 32 lanes of a half wave read 512 contiguous bytes (conflict-free ds_read_b128);
//   * the tile kernel brings a quarter image (24 KB) into LDS with 6 asynchronous global -> LDS copies per thread
//     (no staging registers, no VALU), double-buffered, one barrier per quarter;
//   * A stays fp32 in memory (64 KB per tile, not 96 KB of planes: the kernel streams its tiles from HBM and at
//     2.5-4 us per tile and CU that stream is the next limit), in the A-operand order
//         vals3[tile][w][ks][h][lane][e] = A[32 w + (lane & 31)][16 ks + 8 (lane >> 5) + 4 h + e]
//     (two coalesced float4 loads per lane and k step), and is split in registers next to the MFMAs
//     (v_cvt_pk_bf16_f32 is free beside them);
//   * LDS reads are inline asm with counted lgkmcnt, half a k step ahead (a ds_read the compiler can see makes it
//     wait for ALL outstanding asynchronous copies).
// Structural zeros: as in the fp32 kernel a non-finite sum sends the piece through an exact VALU path (products
// only where A != 0, operands straight from global memory).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pgcn_internal.h"

#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int kT = 128;                  // tile edge
constexpr int kThreads = 256;
constexpr int kQBytes = 3 * 4 * kT * 16; // one quarter image: 3 planes x 4 k groups x 128 columns x 16 B = 24 KB
constexpr int kImgBytes = 4 * kQBytes;   // one panel x feature block: 96 KB
constexpr size_t kSmem3 = 2 * (size_t)kQBytes;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {     // {bf16(x) in bits 0-15, bf16(y) in bits 16-31}, RNE
    const f32x2 v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_as_f32(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi_as_f32(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// x, y -> the three bf16 planes of both, packed {x in bits 0-15, y in bits 16-31}
__device__ __forceinline__ void split_pair(float x, float y, uint32_t &u1, uint32_t &u2, uint32_t &u3) {
    u1 = pack_bf16(x, y);
    const float rx = x - lo_as_f32(u1), ry = y - hi_as_f32(u1);          // exact
    u2 = pack_bf16(rx, ry);
    u3 = pack_bf16(rx - lo_as_f32(u2), ry - hi_as_f32(u2));              // exact, and a bf16 number
}

__device__ __forceinline__ f32x16 mma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- the panel split ------------------------------------------------------------------------------------------
// grid (panels in the list, feature blocks of 128); thread t: column n = t & 127, k groups 8 (t >> 7) .. + 8
__global__ __launch_bounds__(kThreads) void split_panels_kernel(const int32_t *__restrict__ panel_list, const float *__restrict__ B,
                                                                  int64_t ldb, int64_t ncols, int32_t f, u32x4 *__restrict__ image) {
    const int64_t r0 = (int64_t)panel_list[blockIdx.x] * kT;
    const int fcol0 = blockIdx.y * kT;
    const int n = threadIdx.x & (kT - 1);
    const bool n_ok = fcol0 + n < f;
    const float *col = B + fcol0 + (n_ok ? n : 0);
    u32x4 *img = image + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (kImgBytes / 16);
    const int kg0 = (threadIdx.x >> 7) * 8;
#pragma unroll 2
    for (int kgi = 0; kgi < 8; ++kgi) {
        const int kga = kg0 + kgi;                         // k group of the panel: rows 8 kga .. 8 kga + 7
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = r0 + 8 * kga + j;
            x[j] = (n_ok && r < ncols) ? col[r * ldb] : 0.f;
        }
        u32x4 p1, p2, p3;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t u1, u2, u3;
            split_pair(x[2 * d], x[2 * d + 1], u1, u2, u3);
            p1[d] = u1; p2[d] = u2; p3[d] = u3;
        }
        const int q = kga >> 2, kg = kga & 3;
        u32x4 *dst = img + ((q * 3) * 4 + kg) * kT + n;
        dst[0] = p1;
        dst[4 * kT] = p2;
        dst[8 * kT] = p3;
    }
}

// ---- the tile kernel ------------------------------------------------------------------------------------------
// LDS reads the compiler does not see as memory operations; the matching waits take the results as read-write operands
template <int OFF>
__device__ __forceinline__ void lds_read_b128(u32x4 &v, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int N>                                                       // all but the N newest LDS reads have landed
__device__ __forceinline__ void lds_wait(u32x4 (&b)[2][3]) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]) : "n"(N));
}

// the six reads of half step T of a quarter: k step s = T / NH, column blocks 2 h and 2 h + 1 (h = T % NH), three planes
template <int NBLK, int T>
__device__ __forceinline__ void read_half(u32x4 (&bb)[2][3], uint32_t base) {
    constexpr int NH = (NBLK + 1) / 2;
    constexpr int s = T / NH, h = T % NH;
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (2 * h + e < NBLK) {
            lds_read_b128<0 * 8192 + s * 4096 + (2 * h + e) * 512>(bb[e][0], base);
            lds_read_b128<1 * 8192 + s * 4096 + (2 * h + e) * 512>(bb[e][1], base);
            lds_read_b128<2 * 8192 + s * 4096 + (2 * h + e) * 512>(bb[e][2], base);
        }
}

template <int NBLK, int T>
__device__ __forceinline__ void mma_half(f32x16 (&acc)[NBLK], const u32x4 (&a)[3], const u32x4 (&bb)[2][3]) {
    constexpr int NH = (NBLK + 1) / 2;
    constexpr int h = T % NH;
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};     // smallest terms first
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (2 * h + e < NBLK) acc[2 * h + e] = mma(a[pa[i]], bb[e][pb[i]], acc[2 * h + e]);
}

// split this lane's eight A values of one k step (two float4: k = 8 hi + 0..3 and + 4..7) into the three operand planes
__device__ __forceinline__ void split_a(const f32x4 &lo4, const f32x4 &hi4, u32x4 (&a)[3]) {
    split_pair(lo4.x, lo4.y, a[0][0], a[1][0], a[2][0]);
    split_pair(lo4.z, lo4.w, a[0][1], a[1][1], a[2][1]);
    split_pair(hi4.x, hi4.y, a[0][2], a[1][2], a[2][2]);
    split_pair(hi4.z, hi4.w, a[0][3], a[1][3], a[2][3]);
}

// One quarter (32 k) of one tile on the matrix cores: 2 k steps x NH half steps; the B operands of half step t + 1 are
// in flight under the MFMAs of half step t (LDS reads return in order: "all but the reads of half step t + 1").
template <int NBLK>
__device__ __forceinline__ void compute_quarter(f32x16 (&acc)[NBLK], const f32x4 (&af)[4], uint32_t base) {
    constexpr int NH = (NBLK + 1) / 2;
    constexpr int R0 = NBLK >= 2 ? 6 : 3;                 // reads of the first half step of a k step ...
    constexpr int R1 = NBLK == 4 ? 6 : 3;                 // ... and of its second one (NH = 2 only)
    u32x4 b0[2][3] = {}, b1[2][3] = {}, a0[3], a1[3];
    read_half<NBLK, 0>(b0, base);
    split_a(af[0], af[1], a0);
    if constexpr (NH == 1) {
        read_half<NBLK, 1>(b1, base);
        lds_wait<R0>(b0);
        mma_half<NBLK, 0>(acc, a0, b0);
        split_a(af[2], af[3], a1);
        lds_wait<0>(b1);
        mma_half<NBLK, 1>(acc, a1, b1);
    } else {
        read_half<NBLK, 1>(b1, base);
        lds_wait<R1>(b0);
        mma_half<NBLK, 0>(acc, a0, b0);
        read_half<NBLK, 2>(b0, base);
        split_a(af[2], af[3], a1);
        lds_wait<R0>(b1);
        mma_half<NBLK, 1>(acc, a0, b1);
        read_half<NBLK, 3>(b1, base);
        lds_wait<R1>(b0);
        mma_half<NBLK, 2>(acc, a1, b0);
        lds_wait<0>(b1);
        mma_half<NBLK, 3>(acc, a1, b1);
    }
}

// the 6 asynchronous copies of one quarter image per thread (wave-uniform LDS destination, lane-linear image)
__device__ __forceinline__ void issue_quarter(const char *__restrict__ src, char *smem, int buf, int w) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 4096 + (int)threadIdx.x * 16),
                                         (lptr_t)(smem + buf * kQBytes + i * 4096 + w * 1024), 16, 0, 0);
}

template <int NBLK>
__device__ __forceinline__ void dense3_piece(const int4 wk, const int32_t *__restrict__ tile_img, const f32x4 *__restrict__ vals3,
                                             const char *__restrict__ image, int nfb, int fb, char *smem, f32x16 (&acc)[NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t rbase = lds0 + hi * 2048 + lo * 16;
    auto img_of = [&](int qi) -> const char * {          // quarter qi of the piece: tile wk.y + qi / 4, quarter qi % 4
        const int64_t pi = tile_img[(int64_t)wk.y + (qi >> 2)];
        return image + (pi * nfb + fb) * (int64_t)kImgBytes + (qi & 3) * kQBytes;
    };
    auto load_a = [&](int qi, f32x4 (&af)[4]) {          // k steps 2 (qi % 4) and + 1 of the tile: [s][h]
        const f32x4 *ap = vals3 + ((((int64_t)wk.y + (qi >> 2)) * 4 + w) * 8 + 2 * (qi & 3)) * 2 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = ap[i * 64];
    };
    const int nq = wk.z * 4;                              // even, >= 4
    f32x4 afA[4], afB[4];
    issue_quarter(img_of(0), smem, 0, w);
    load_a(0, afA);
    for (int it = 0; it < nq; it += 2) {
        // quarter it: buffer 0, operands afA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's copies of quarter `it` (and its A operands) have landed
        __syncthreads();                                   // ... everybody's have; nobody reads buffer 1 (quarter it - 1) any more
        issue_quarter(img_of(it + 1), smem, 1, w);
        load_a(it + 1, afB);
        compute_quarter<NBLK>(acc, afA, rbase);
        // quarter it + 1: buffer 1, operands afB
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (it + 2 < nq) {
            issue_quarter(img_of(it + 2), smem, 0, w);
            load_a(it + 2, afA);
        }
        compute_quarter<NBLK>(acc, afB, rbase + kQBytes);
    }
}

// Exact redo of a piece: products only where A != 0, k ascending, operands from global memory.
template <int NBLK>
__device__ __noinline__ void dense3_piece_exact(const int4 wk, const int32_t *__restrict__ tile_img,
                                                const int32_t *__restrict__ panel_list, const float *__restrict__ vals3,
                                                const float *__restrict__ B, int64_t ldb, int64_t ncols, int fcol0, int fw,
                                                f32x16 (&acc)[NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    for (int t = 0; t < wk.z; ++t) {
        const int64_t ti = (int64_t)wk.y + t;
        const int64_t prow0 = (int64_t)panel_list[tile_img[ti]] * kT;
        for (int k = 0; k < kT; ++k) {
            const int ks = k >> 4, hk = (k >> 3) & 1, h = (k >> 2) & 1, e = k & 3;
            float b[NBLK];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const int colj = nb * 32 + lo;
                b[nb] = (prow0 + k < ncols && colj < fw) ? B[(prow0 + k) * ldb + fcol0 + colj] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float x = vals3[(((((ti * 4 + w) * 8 + ks) * 2 + h) * 64) + hk * 32 + il) * 4 + e];
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) acc[nb][r] = x != 0.f ? fmaf(x, b[nb], acc[nb][r]) : acc[nb][r];
            }
        }
    }
}

// work: int4 {tile row, first tile, number of tiles, first slot}; NBLK = 32-column blocks holding features
template <int NBLK>
__global__ __launch_bounds__(kThreads, 2) void spmm_dense3_kernel(
    const int4 *__restrict__ work, const int32_t *__restrict__ tile_img, const int32_t *__restrict__ panel_list,
    const float *__restrict__ vals3, const char *__restrict__ image, const float *__restrict__ B, int64_t ldb, int64_t ncols,
    int32_t f, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    int4 wk = work[blockIdx.x];
    wk.y = __builtin_amdgcn_readfirstlane(wk.y); wk.z = __builtin_amdgcn_readfirstlane(wk.z);
    const int fb = blockIdx.y, nfb = gridDim.y;
    const int fcol0 = fb * kT;
    const int fw = min(kT, f - fcol0);
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    dense3_piece<NBLK>(wk, tile_img, reinterpret_cast<const f32x4 *>(vals3), image, nfb, fb, smem3, acc);
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
        dense3_piece_exact<NBLK>(wk, tile_img, panel_list, vals3, B, ldb, ncols, fcol0, fw, exact);
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

extern "C" int64_t pgcn_dense_bf16x3_image_bytes(int64_t npanels, int32_t f) {
    return npanels * (int64_t)((f + kT - 1) / kT) * kImgBytes;
}

extern "C" int pgcn_spmm_dense_bf16x3_f32(const int32_t *work, int64_t nwork, const int32_t *tile_img, const float *vals3,
                                          const int32_t *panel_list, int64_t npanels, const float *B, int64_t ldb,
                                          int64_t ncols, int32_t f, void *image_ws, int64_t image_ws_bytes,
                                          float *partial_ws, int64_t partial_ws_elems, int64_t nslots_total,
                                          pgcn_stream_t stream) {
    if (nwork < 0 || npanels < 0 || f <= 0 || ldb < f || ncols < 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: bad sizes");
    if (nwork == 0) return PGCN_OK;
    if (!work || !tile_img || !vals3 || !panel_list || !B || !image_ws || !partial_ws || npanels == 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: null pointer");
    if ((uintptr_t)vals3 % 16 || (uintptr_t)image_ws % 16 || (uintptr_t)work % 16)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: work / vals3 / image_ws must be 16-byte aligned");
    if (image_ws_bytes < pgcn_dense_bf16x3_image_bytes(npanels, f))
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_dense_bf16x3_f32: panel image work-space too small");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_dense_bf16x3_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL || npanels > 0x7fffffffLL)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: work / panel list too long");
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
    hipStream_t s = (hipStream_t)stream;
    const unsigned nfb = (unsigned)((f + kT - 1) / kT);
    hipLaunchKernelGGL(split_panels_kernel, dim3((unsigned)npanels, nfb), dim3(kThreads), 0, s, panel_list, B, ldb, ncols, f,
                       reinterpret_cast<u32x4 *>(image_ws));
    PGCN_HIP_CHECK(hipGetLastError());
    const int4 *w4 = reinterpret_cast<const int4 *>(work);
    const char *img = reinterpret_cast<const char *>(image_ws);
    const dim3 grid((unsigned)nwork, nfb), block(kThreads);
    switch (((f < kT ? f : kT) + 31) / 32) {
        case 1: hipLaunchKernelGGL(spmm_dense3_kernel<1>, grid, block, kSmem3, s, w4, tile_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
        case 2: hipLaunchKernelGGL(spmm_dense3_kernel<2>, grid, block, kSmem3, s, w4, tile_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
        case 3: hipLaunchKernelGGL(spmm_dense3_kernel<3>, grid, block, kSmem3, s, w4, tile_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
        default: hipLaunchKernelGGL(spmm_dense3_kernel<4>, grid, block, kSmem3, s, w4, tile_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
