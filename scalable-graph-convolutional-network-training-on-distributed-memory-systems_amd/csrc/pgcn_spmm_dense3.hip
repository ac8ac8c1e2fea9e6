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
// What the r03 harness version taught (tools/experiments/dense3, 7.9 us per 128 x 128 tile and CU against 10.9 for
// fp32): splitting the feature panel INSIDE the tile loop costs more than the MFMAs (per quarter panel and wave: 2 075
// ticks to issue the dword loads + 1 070 to split and write LDS against 1 536 matrix-pipe cycles), and it is redone
// by every tile that shares the panel.  And what the first r04 version taught (panels split once per SpMM, 128-row
// tiles, two workgroups per CU: 7.7 us per tile): the kernel is then bound by the L2 <-> fabric traffic -- 96 KB of
// planes + 64 KB of A per 128 x 128 tile at the ~21 GB/s a CU gets when all 256 pull (5.4 TB/s in total).  So:
//   * spmm_split_panels_kernel splits the panels that dense blocks refer to ONCE per SpMM (~490 of 1 821 panels on the
//     benchmark graph: 31 MB read, 47 MB written) into the exact LDS image of the block kernel, in a work-space:
//         image[panel][feature block] = [quarter q (32 k)][plane p][k group kg (8 k)][column n (128)] x 16 B
//     (8 bf16: k = 32 q + 8 kg + j), i.e. a lane's B operand of v_mfma_f32_32x32x16_bf16 is one 16-byte slot and
//     the 32 lanes of a half wave read 512 contiguous bytes (conflict-free ds_read_b128);
//   * a BLOCK is 512 rows x 128 columns (the row blocking of the strip tiles: the dense corner of a degree-sorted
//     power-law graph is a nest of rectangles, 93 % of its 128 x 128 tiles sit in such blocks): one workgroup of
//     8 waves, wave w = rows [64 w, 64 w + 64) x 128 features (2 x 4 accumulator blocks), so a panel image is
//     fetched once per FOUR 128 x 128 tiles (24 + 64 KB per tile instead of 96 + 64) and every B operand read
//     from LDS feeds two MFMAs;
//   * the kernel brings a quarter image (24 KB) into LDS with 3 asynchronous global -> LDS copies per thread
//     (no staging registers, no VALU), double-buffered, one barrier per quarter;
//   * A stays fp32 in memory, in the A-operand order
//         vals3[block][w][unit = 2 ks + rb][h][lane][e] = A[64 w + 32 rb + (lane & 31)][16 ks + 8 (lane >> 5) + 4 h + e]
//     (64 KB per 128 x 128 of the block; 2 KB per wave and unit), and is split in registers.  Pre-split planes
//     (96 KB per 128 x 128, no VALU work) were built and measured too: the kernel then moves 120 KB per tile and
//     is bound by it (6.2 us per tile with the MFMAs removed, 7.7 with them; HBM + fabric deliver ~5.8 TB/s).
//     The split costs 44 VALU instructions per 24 MFMAs, and WHERE they sit decides what they cost: inside an
//     accumulator's chain of six MFMAs every clump costs the chain its forwarding path (+43 cycles: 6.7 us per
//     tile against 4.6 with the split removed), so they sit between the chains, one pair of values per step,
//     where the other wave of the SIMD owns the matrix pipe anyway;
//   * every wave brings its own A values into a wave-private LDS ring (2 asynchronous copies of 1 KB per unit,
//     four units = one quarter ahead: no staging registers, no barrier), reads them two units ahead and splits
//     them one unit ahead;
//   * LDS reads are inline asm with counted lgkmcnt, half a k step ahead (a ds_read the compiler can see makes it
//     wait for ALL outstanding asynchronous copies).
// Structural zeros: as in the fp32 kernel a non-finite sum sends the piece through an exact VALU path (products
// only where A != 0, operands straight from global memory).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "pgcn_bf16x3.h"
#include "pgcn_internal.h"
#include "pgcn_once.h"

#pragma clang diagnostic ignored "-Winline-asm"

// Measurement builds (r04's tools/micro/dense3_bench, in the git history, compiled this file again with -DPGCN_DENSE3_PROBE=N and another
// entry-point name; the library is built with 0 and every branch below folds away).  1: no MFMAs (TIMING ONLY, wrong
// sums); 5: the real kernel with s_memtime phase timers: every wave writes {ticks waiting at the top of a quarter
// (copies + barrier), ticks computing, quarters, whole loop} to the first float of its first four slot rows.
#ifndef PGCN_DENSE3_PROBE
#define PGCN_DENSE3_PROBE 0
#endif

namespace {
using namespace pgcn_bf16x3;

constexpr int kProbe = PGCN_DENSE3_PROBE;
constexpr int kBR = PGCN_STRIP_TR;       // 512 rows per block
constexpr int kThreads = 512;            // 8 waves x 64 rows
static_assert(kBR == 8 * 64, "a wave owns 64 rows of a block");
constexpr int kUnitBytes = 2 * 1024;     // a wave's A values of one unit (k step x row block): 2 float4 per lane
constexpr int kRingUnits = 4;            // units of A in flight per wave (one quarter)
constexpr int kOffA = 2 * kQBytes;       // LDS: two quarter images of B, then the A rings [slot][wave]
constexpr size_t kSmem3 = kOffA + (size_t)kRingUnits * 8 * kUnitBytes;
static_assert(kSmem3 <= 160 * 1024, "LDS budget of one CU");

__device__ __forceinline__ f32x16 mma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    if constexpr (kProbe == 1) return c;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- the tile kernel ------------------------------------------------------------------------------------------
// (LDS reads: pgcn_bf16x3.h lds_read_b128 -- inline asm the compiler does not see; the matching waits take the results as read-write operands)
template <int N>                                                       // all but the N newest LDS reads have landed
__device__ __forceinline__ void lds_wait(u32x4 (&b)[3]) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N));
}

// the three reads (one per plane) of step T of a quarter: k step s = T / NBLK, column block nb = T % NBLK
template <int NBLK, int T>
__device__ __forceinline__ void read_step(u32x4 (&bb)[3], uint32_t base) {
    constexpr int s = T / NBLK, nb = T % NBLK;
    lds_read_b128<0 * 8192 + s * 4096 + nb * 512>(bb[0], base);
    lds_read_b128<1 * 8192 + s * 4096 + nb * 512>(bb[1], base);
    lds_read_b128<2 * 8192 + s * 4096 + nb * 512>(bb[2], base);
}

// six MFMAs of one (k step, row block, column block): acc[rb][nb] += A[rb] . B[nb], smallest terms first -- one
// uninterrupted chain on one accumulator (anything between two MFMAs of a chain costs its forwarding path)
template <int NBLK, int RB, int NB>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[2][NBLK], const u32x4 (&a)[3], const u32x4 (&bb)[3]) {
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[RB][NB] = mma(a[pa[i]], bb[pb[i]], acc[RB][NB]);
}

// pair D (0..3) of the eight A values of a lane and unit: D = 0, 1 -> lo4.xy, lo4.zw (k = 8 hi + 0..3); 2, 3 -> hi4.xy, hi4.zw
template <int D>
__device__ __forceinline__ void split_pair_of(const f32x4 &lo4, const f32x4 &hi4, u32x4 (&a)[3]) {
    float x, y;
    if constexpr (D == 0) { x = lo4.x; y = lo4.y; }
    else if constexpr (D == 1) { x = lo4.z; y = lo4.w; }
    else if constexpr (D == 2) { x = hi4.x; y = hi4.y; }
    else { x = hi4.z; y = hi4.w; }
    uint32_t u1, u2, u3;
    split_pair(x, y, u1, u2, u3);
    a[0][D] = u1; a[1][D] = u2; a[2][D] = u3;
}

template <int N>                                                       // ... and the same with the two registers of fresh A values
__device__ __forceinline__ void lds_wait2(u32x4 (&b)[3], f32x4 (&a)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(a[0]), "+v"(a[1]) : "n"(N));
}

// the 3 asynchronous copies of one quarter image of B per thread (wave-uniform LDS destination, lane-linear image)
__device__ __forceinline__ void issue_quarter(const char *__restrict__ src, char *smem, int buf, int w) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + i * 8192 + (int)threadIdx.x * 16),
                                         (lptr_t)(smem + buf * kQBytes + i * 8192 + w * 1024), 16, 0, 0);
}
// the 2 asynchronous copies of this wave's A values of one unit into its ring slot `slot` ([wave][slot] x 2 KB)
__device__ __forceinline__ void issue_unit(const char *__restrict__ src_lane, char *smem, int slot, int w) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
        __builtin_amdgcn_global_load_lds((gptr_t)(src_lane + h * 1024),
                                         (lptr_t)(smem + kOffA + (w * kRingUnits + slot) * kUnitBytes + h * 1024), 16, 0, 0);
}

// One quarter (32 k) of one block on the matrix cores, as four UNITS u = 2 s + rb (k step s, row block rb) of NBLK steps
// of six MFMAs.  A travels ring -> registers -> planes:  in unit u the fp32 values of unit u + 2 are read from ring slot
// v = (u + 2) % 4 into af[u & 1] (and, once those reads have returned, the same unit ONE QUARTER ON is requested into
// the slot), while the values of unit u + 1 (af[(u + 1) & 1], read one unit ago) are split into the plane set
// ap[(u + 1) & 1], one pair per step, BETWEEN the chains.  The B operands of step t + 1 are in flight under the MFMAs
// of step t (LDS reads return in order: "all but the newest").  a1 / a2: this lane's address of unit 0 of quarter
// q + 1 / q + 2.
//
// Copies of a wave in issue order:  B(q+1) [3 copies] at the top of quarter q, then one group of A [2 copies] per unit:
// A(q+1,2), A(q+1,3), A(q+2,0), A(q+2,1) in units 0..3 of quarter q.  What a unit reads was requested four units
// earlier -- three A groups and one B group are newer: vmcnt(9); B(q) at the top of quarter q has four A groups
// behind it: vmcnt(8).  (Copies complete in issue order, and every copy has a full quarter of MFMAs to land.)
template <int NBLK>
__device__ __forceinline__ void compute_quarter(f32x16 (&acc)[2][NBLK], u32x4 (&ap)[2][3], f32x4 (&af)[2][2], uint32_t base,
                                                uint32_t abase, const char *__restrict__ a1, const char *__restrict__ a2,
                                                char *smem, int w) {
    constexpr int NT = 4 * NBLK;                          // steps t = u * NBLK + nb
    constexpr int NBC = NBLK == 1 ? 0 : 1;                // the step of a unit whose wait confirms the A reads
    u32x4 b[2][3];
    read_step<NBLK, 0>(b[0], base);
    static_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int u = t / NBLK, nb = t % NBLK, rb = u % 2;
        constexpr int v = (u + 2) % 4;
        if constexpr (nb == 0) {
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            lds_read_b128<v * kUnitBytes + 0>(af[u & 1][0], abase);
            lds_read_b128<v * kUnitBytes + 1024>(af[u & 1][1], abase);
        }
        if constexpr (t + 1 < NT) {
            constexpr int u1 = (t + 1) / NBLK, nb1 = (t + 1) % NBLK;
            read_step<NBLK, (u1 / 2) * NBLK + nb1>(b[(t + 1) & 1], base);      // (both row blocks of a k step read the same operands)
            if constexpr (nb == NBC) lds_wait2<3>(b[t & 1], af[u & 1]);         // ... the A reads of this unit have returned too
            else if constexpr (nb == 0) lds_wait<5>(b[t & 1]);                  // (the two A reads are newer)
            else lds_wait<3>(b[t & 1]);
        } else {
            lds_wait2<0>(b[t & 1], af[u & 1]);
        }
        if constexpr (nb == NBC) issue_unit((u < 2 ? a1 : a2) + v * kUnitBytes, smem, v, w);   // the slot is free: the same unit one quarter on
        // the split of unit u + 1: pair nb under this step (a narrow operand has fewer steps than pairs: the rest under the last one)
        static_for<nb, (nb == NBLK - 1 ? 4 : nb + 1)>([&](auto dc) {
            split_pair_of<decltype(dc)::value>(af[(u + 1) & 1][0], af[(u + 1) & 1][1], ap[(u + 1) & 1]);
        });
        __builtin_amdgcn_sched_barrier(0);                // (nothing inside the chain of six)
        mma_step<NBLK, rb, nb>(acc, ap[u & 1], b[t & 1]);
        // (left alone the scheduler sinks the MFMAs of a whole unit below the reads and waits of its later steps --
        //  which then wait for the LDS with nothing in the matrix pipe)
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <int NBLK>
__device__ __forceinline__ void dense3_piece(const int4 wk, const int32_t *__restrict__ blk_img, const char *__restrict__ vals3,
                                             const char *__restrict__ image, int nfb, int fb, char *smem, f32x16 (&acc)[2][NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t rbase = lds0 + hi * 2048 + lo * 16;
    const uint32_t abase = lds0 + kOffA + w * kRingUnits * kUnitBytes + lane * 16;
    const int nq = wk.z * 4;
    auto img_of = [&](int qi) -> const char * {          // quarter qi of the piece (clamped): block wk.y + qi / 4, quarter qi % 4
        qi = qi < nq ? qi : nq - 1;                       // (the last quarters fetch themselves again instead of branching)
        const int64_t pi = blk_img[(int64_t)wk.y + (qi >> 2)];
        return image + (pi * nfb + fb) * (int64_t)kImgBytes + (qi & 3) * kQBytes;
    };
    // this lane's address of unit 0 of quarter qi (clamped): vals3[block][w][16 units][2][64 lanes] x 16 B
    auto a_of = [&](int qi) -> const char * {
        qi = qi < nq ? qi : nq - 1;
        return vals3 + (((((int64_t)wk.y + (qi >> 2)) * 8 + w) * 16 + 4 * (qi & 3)) * 2 * 64 + lane) * 16;
    };
    u32x4 ap[2][3];                                       // bf16 planes of the current and the next unit
    f32x4 af[2][2];                                       // fp32 values of the next unit and the one after
    // prologue: build the steady state of the top of quarter 0 -- the planes of unit (0,0) in ap[0], the values of unit
    // (0,1) in af[1]; in flight, oldest first: B(0), A(0,2), A(0,3), A(1,0), A(1,1)
    issue_unit(a_of(0), smem, 0, w);
    issue_unit(a_of(0) + kUnitBytes, smem, 1, w);
    issue_quarter(img_of(0), smem, 0, w);
    issue_unit(a_of(0) + 2 * kUnitBytes, smem, 2, w);
    issue_unit(a_of(0) + 3 * kUnitBytes, smem, 3, w);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");      // A(0,0) and A(0,1) have landed
    lds_read_b128<0>(af[0][0], abase);
    lds_read_b128<1024>(af[0][1], abase);
    lds_read_b128<kUnitBytes>(af[1][0], abase);
    lds_read_b128<kUnitBytes + 1024>(af[1][1], abase);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]));
    static_for<0, 4>([&](auto dc) { split_pair_of<decltype(dc)::value>(af[0][0], af[0][1], ap[0]); });
    issue_unit(a_of(1), smem, 0, w);
    issue_unit(a_of(1) + kUnitBytes, smem, 1, w);
    long long t_top = 0, t_cmp = 0, t0 = 0, t1 = 0, t_begin = 0;
    auto tick = [&]() -> long long {
        if constexpr (kProbe == 5) {
            __builtin_amdgcn_sched_barrier(0);
            const long long t = (long long)__builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            return t;
        } else {
            return 0;
        }
    };
    t_begin = t0 = tick();
    for (int q = 0; q < nq; ++q) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's part of B(q) has landed
        __syncthreads();                                   // ... everybody's has; nobody reads the other buffer (quarter q - 1) any more
        t1 = tick(); t_top += t1 - t0;
        issue_quarter(img_of(q + 1), smem, (q + 1) & 1, w);
        compute_quarter<NBLK>(acc, ap, af, rbase + (q & 1) * kQBytes, abase, a_of(q + 1), a_of(q + 2), smem, w);
        t0 = tick(); t_cmp += t0 - t1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the redundant last fetches must not outlive the workgroup's LDS)
    if constexpr (kProbe == 5) {
        if (lane == 0) {
            acc[0][0][0] = (float)t_top; acc[0][0][1] = (float)t_cmp; acc[0][0][2] = (float)nq; acc[0][0][3] = (float)(t0 - t_begin);
        }
    }
}

// Exact redo of a piece: products only where A != 0, k ascending, operands from global memory.
template <int NBLK>
__device__ __noinline__ void dense3_piece_exact(const int4 wk, const int32_t *__restrict__ blk_img,
                                                const int32_t *__restrict__ panel_list, const float *__restrict__ vals3,
                                                const float *__restrict__ B, int64_t ldb, int64_t ncols, int fcol0, int fw,
                                                f32x16 (&acc)[2][NBLK]) {
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int hi = lane >> 5, lo = lane & 31;
    for (int t = 0; t < wk.z; ++t) {
        const int64_t bi = (int64_t)wk.y + t;
        const int64_t prow0 = (int64_t)panel_list[blk_img[bi]];
        for (int k = 0; k < kT; ++k) {
            const int ks = k >> 4, hk = (k >> 3) & 1, h = (k >> 2) & 1, e = k & 3;
            float b[NBLK];
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const int colj = nb * 32 + lo;
                b[nb] = (prow0 + k < ncols && colj < fw) ? B[(prow0 + k) * ldb + fcol0 + colj] : 0.f;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float x = vals3[((((((bi * 8 + w) * 16 + 2 * ks + rb) * 2 + h) * 64) + hk * 32 + il) * 4) + e];
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb) acc[rb][nb][r] = x != 0.f ? fmaf(x, b[nb], acc[rb][nb][r]) : acc[rb][nb][r];
                }
        }
    }
}

// work: int4 {block row, first block, number of blocks, first slot}; NBLK = 32-column blocks holding features
template <int NBLK>
__global__ __launch_bounds__(kThreads, 2) void spmm_dense3_kernel(
    const int4 *__restrict__ work, const int32_t *__restrict__ blk_img, const int32_t *__restrict__ panel_list,
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
    f32x16 acc[2][NBLK];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.f;
    dense3_piece<NBLK>(wk, blk_img, reinterpret_cast<const char *>(vals3), image, nfb, fb, smem3, acc);
    bool bad = false;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) bad = bad || !(fabsf(acc[rb][nb][r]) <= 3.402823466e+38f);
    if (__syncthreads_or(bad)) {
        f32x16 exact[2][NBLK];   // (its own array: the address of `acc` must not escape, or the accumulators live in scratch)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) exact[rb][nb][r] = 0.f;
        dense3_piece_exact<NBLK>(wk, blk_img, panel_list, vals3, B, ldb, ncols, fcol0, fw, exact);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) acc[rb][nb] = exact[rb][nb];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const int colj = nb * 32 + lo;
            if (colj < fw) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    partial[((int64_t)wk.w + 64 * w + 32 * rb + il) * f + fcol0 + colj] = acc[rb][nb][r];
                }
            }
        }
}

}  // namespace

extern "C" int64_t pgcn_dense_bf16x3_image_bytes(int64_t npanels, int32_t f) {
    return npanels * (int64_t)((f + kT - 1) / kT) * kImgBytes;
}

extern "C" int pgcn_spmm_dense_bf16x3_f32(const int32_t *work, int64_t nwork, const int32_t *blk_img, const float *vals3,
                                          const int32_t *panel_list, int64_t npanels, const float *B, int64_t ldb,
                                          int64_t ncols, int32_t f, void *image_ws, int64_t image_ws_bytes,
                                          float *partial_ws, int64_t partial_ws_elems, int64_t nslots_total,
                                          pgcn_stream_t stream) {
    if (nwork < 0 || npanels < 0 || f <= 0 || ldb < f || ncols < 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: bad sizes");
    if (nwork == 0) return PGCN_OK;
    if (!work || !blk_img || !vals3 || !panel_list || !B || !image_ws || !partial_ws || npanels == 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: null pointer");
    if ((uintptr_t)vals3 % 16 || (uintptr_t)image_ws % 16 || (uintptr_t)work % 16)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: work / vals3 / image_ws must be 16-byte aligned");
    if (image_ws_bytes < pgcn_dense_bf16x3_image_bytes(npanels, f))
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_dense_bf16x3_f32: panel image work-space too small");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_dense_bf16x3_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL || npanels > 0x7fffffffLL)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_dense_bf16x3_f32: work / panel list too long");
    static PgcnPerDeviceOnce once;
    if (int rc = once.run([&]() -> int {
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_dense3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem3));
            return PGCN_OK;
        }))
        return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned nfb = (unsigned)((f + kT - 1) / kT);
    hipLaunchKernelGGL(spmm_split_panels_kernel, dim3((unsigned)npanels, nfb), dim3(kSplitThreads), 0, s, panel_list, B, ldb, ncols, f,
                       reinterpret_cast<u32x4 *>(image_ws));
    PGCN_HIP_CHECK(hipGetLastError());
    const int4 *w4 = reinterpret_cast<const int4 *>(work);
    const char *img = reinterpret_cast<const char *>(image_ws);
    const dim3 grid((unsigned)nwork, nfb), block(kThreads);
    switch (((f < kT ? f : kT) + 31) / 32) {
        case 1: hipLaunchKernelGGL(spmm_dense3_kernel<1>, grid, block, kSmem3, s, w4, blk_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
        case 2: hipLaunchKernelGGL(spmm_dense3_kernel<2>, grid, block, kSmem3, s, w4, blk_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
        case 3: hipLaunchKernelGGL(spmm_dense3_kernel<3>, grid, block, kSmem3, s, w4, blk_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
        default: hipLaunchKernelGGL(spmm_dense3_kernel<4>, grid, block, kSmem3, s, w4, blk_img, panel_list, vals3, img, B, ldb, ncols, f, partial_ws); break;
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
