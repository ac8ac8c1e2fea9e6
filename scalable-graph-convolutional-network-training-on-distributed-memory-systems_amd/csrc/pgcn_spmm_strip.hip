// pgcn_spmm_strip.hip -- LDS-staged SpMM over TALL tiles (512 rows x 128 columns) for gfx950.
//
// Why.  A gather through the vector L1 moves one 512 B row of the dense operand per stored entry
// and saturates near 18 TB/s on an MI355X even when every row hits in L2 (r02 probe, uniform hot
// set): at 58.8 GB of gathers per Reddit-sized SpMM that is the whole budget.  Staging a 128-row
// panel of B once in LDS (64 KB through the same L1 path) and serving all entries of a 512-row
// tile from LDS (256 B/clk/CU) replaces 512 B per ENTRY by 64 KB per TILE: it pays from 128
// entries per tile (0.2 % fill) on, and on the degree-sorted benchmark graph cuts the bytes that
// cross the L1 from 21 GB to 7 GB (part of /root/reference/GPU/PGCN.py:127's torch.sparse.mm).
//
// Shape.  One workgroup = 1024 threads = 32 groups of 32 lanes; a lane owns 4 consecutive
// features (float4), a group one row at a time, 16 row slots per group: local row = j * 32 + g,
// 16 x float4 accumulators per lane stay in registers over the whole piece (a run of records of
// one tile row).  A RECORD is one LAYER of a tile: the (2 l)-th and (2 l + 1)-th stored entry of
// every row, i.e. exactly 2 (offset, value) pairs per row = 8 KB.  A first version walked
// per-row counted segments; with ~1.3 entries per row and panel its loop overhead cost 29 clk per
// entry (VALU issue bound).  With a fixed shape the compute phase is straight-line: 16 pair reads,
// 32 row reads, 64 packed FMAs per lane and record, no branch, no count, no address arithmetic
// beyond one add (the pair holds the row's byte offset inside the staged panel).  An unused slot
// points at an all-zero LDS row with value 0.0: no row the matrix does not reference is ever
// combined (no 0 x Inf).
//
// Pipeline.  The panel (4 x 16 B per thread) and the pairs (8 KB) of a record arrive by
// asynchronous global -> LDS copies (global_load_lds_dwordx4) into double buffers, issued one
// record ahead.  The compute phase reads LDS through inline asm: a ds_read the compiler can see
// makes it wait for ALL outstanding copies (vmcnt(0)), i.e. for the prefetch as well; raw
// s_barrier + counted vmcnt keep the next record's copies in flight across the barriers.  Record
// headers come by scalar loads.  Deterministic: fixed order, no atomics; a piece writes 512
// partial rows to slots that pgcn_spmm_fixup_f32 adds in list order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pgcn_internal.h"

namespace {

constexpr int TRS = PGCN_STRIP_TR;     // 512 rows per strip tile
constexpr int TC = PGCN_CORE_TC;       // 128 columns per panel
constexpr int SB = PGCN_STRIP_B;       // pair slots per row and record
constexpr int kThreads = 1024;
constexpr int NG = kThreads / 32;      // 32 groups
constexpr int RW = TRS / NG;           // 16 row slots per group
static_assert(RW == 16 && NG == 32 && SB == 2, "layout constants are baked into the record format");

constexpr int kPanelBytes = (TC + 1) * 512;           // 128 rows x 128 fp32 + the all-zero row
constexpr int kPadOff = TC * 512;                     // what an unused pair slot points at
constexpr int kRecBytes = TRS * SB * 8;               // 8 KB of pairs per record
constexpr int kOffPairs = 2 * kPanelBytes;
constexpr int kSmem = kOffPairs + 2 * kRecBytes;
static_assert(kSmem <= 160 * 1024, "LDS budget of one CU");
static_assert(kRecBytes == 8 * 64 * 16, "pairs are copied by waves 0..7, one 16-byte piece per lane");

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// LDS reads the compiler does not see as memory operations; the matching wait takes the results
// as read-write operands so that every use is ordered behind it.  LDS returns data in order, so
// lgkmcnt(N) leaves the N youngest reads in flight.
template <int OFF>
__device__ __forceinline__ void lds_read_b128(f32x4 &v, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x4 &a, f32x4 &b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// Asynchronous copies of one record: 4 per thread for a staged panel + 1 for waves 0..7 (pairs).
__device__ __forceinline__ void issue_record(int panel, bool stage, char *smem, int pb, int qb,
                                             const float *__restrict__ B, int64_t ldb, int64_t ncols, int fcol0,
                                             int fw, const int32_t *__restrict__ pairs, int64_t k, int wave, int lane) {
    if (stage) {
        const int64_t col0 = (int64_t)panel * TC;
        const int lastc4 = (fw >> 2) - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = q * kThreads + wave * 64 + lane;
            const int row = idx >> 5;
            int c4 = idx & 31;
            c4 = c4 < lastc4 ? c4 : lastc4;                                   // narrow panels: repeat the last vector
            int64_t gr = col0 + row;
            gr = gr < ncols ? gr : ncols - 1;                                 // rows past the operand: never referenced
            __builtin_amdgcn_global_load_lds((gptr_t)(B + gr * ldb + fcol0 + c4 * 4),
                                             (lptr_t)(smem + pb * kPanelBytes + (q * kThreads + wave * 64) * 16), 16, 0, 0);
        }
    }
    if (wave < 8)
        __builtin_amdgcn_global_load_lds((gptr_t)(pairs + k * (int64_t)(TRS * SB * 2) + (wave * 64 + lane) * 4),
                                         (lptr_t)(smem + kOffPairs + qb * kRecBytes + wave * 64 * 16), 16, 0, 0);
}

__device__ __forceinline__ void wait_for_previous(bool next_staged, bool extra) {
    // the copies of the NEXT record stay in flight: 4 (panel) + 1 (pairs) per thread
    if (next_staged) { if (extra) wait_vm<5>(); else wait_vm<4>(); }
    else             { if (extra) wait_vm<1>(); else wait_vm<0>(); }
}

__device__ __forceinline__ void fma_row(f32x2 (&acc)[2], float w, const f32x4 &x) {
    const f32x2 ww = {w, w};
    acc[0] = __builtin_elementwise_fma(ww, f32x2{x.x, x.y}, acc[0]);
    acc[1] = __builtin_elementwise_fma(ww, f32x2{x.z, x.w}, acc[1]);
}

// work: int4 {tile row, first record, one-past-last record, first slot}; recs: int4 {panel, flags, -, -}
// PROBE: 0 = the product kernel; 1 = no compute phase, 2 = no panel staging (compile-time variants used once to
// separate the LDS-bound compute from the staging pipeline: DESIGN.md 4; selected by PGCN_STRIP_PROBE)
template <int PROBE>
__global__ __launch_bounds__(kThreads, 1) void spmm_strip_kernel(
    const int4 *__restrict__ work, const int4 *__restrict__ recs, const int32_t *__restrict__ pairs,
    const float *__restrict__ B, int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int4 wk = work[blockIdx.x];
    const int fcol0 = blockIdx.y * 128;
    const int fw = min(128, f - fcol0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane & 31;
    const int group = threadIdx.x >> 5;
    const bool extra = wave < 8;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    f32x2 acc[RW][2];
#pragma unroll
    for (int j = 0; j < RW; ++j) acc[j][0] = acc[j][1] = f32x2{0.f, 0.f};
    if (threadIdx.x < 64)   // the all-zero row of both panel buffers
        *reinterpret_cast<float4 *>(smem + (threadIdx.x >> 5) * kPanelBytes + kPadOff + (threadIdx.x & 31) * 16) =
            make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... has left this wave before the first barrier

    int pb = 0;
    issue_record(recs[wk.y].x, true, smem, 0, 0, B, ldb, ncols, fcol0, fw, pairs, wk.y, wave, lane);
    for (int k = wk.y; k < wk.z; ++k) {
        const int qb = (k - wk.y) & 1;
        int pbn = pb;
        if (k + 1 < wk.z) {
            const int4 nx = recs[k + 1];
            const bool stage = (nx.y & 1) == 0 && !(PROBE & 2);
            pbn = stage ? pb ^ 1 : pb;
            issue_record(nx.x, stage, smem, pbn, qb ^ 1, B, ldb, ncols, fcol0, fw, pairs, k + 1, wave, lane);
            wait_for_previous(stage, extra);
        } else {
            wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();      // every thread's copies of record k have landed
        if constexpr (!(PROBE & 1)) {
            const uint32_t pa = lds0 + kOffPairs + qb * kRecBytes + group * (RW * SB * 8);   // this group's 16 x 2 pairs
            const uint32_t rowbase = lds0 + pb * kPanelBytes + sub * 16;
            // software pipeline over pairs of row slots: the pair reads of slots J+2, J+3 are in flight
            // behind the four row reads of slots J, J+1 (LDS returns in order: lgkmcnt(2) = rows landed)
            f32x4 p0, p1, n0, n1;
            lds_read_b128<0>(p0, pa);
            lds_read_b128<16>(p1, pa);
#define PGCN_STRIP_STEP(J, MORE)                                                        \
            {                                                                           \
                lds_wait<0>(p0, p1);                                                    \
                f32x4 x0, x1, x2, x3;                                                   \
                lds_read_b128<0>(x0, rowbase + (uint32_t)__float_as_int(p0.x));         \
                lds_read_b128<0>(x1, rowbase + (uint32_t)__float_as_int(p0.z));         \
                lds_read_b128<0>(x2, rowbase + (uint32_t)__float_as_int(p1.x));         \
                lds_read_b128<0>(x3, rowbase + (uint32_t)__float_as_int(p1.z));         \
                if (MORE) {                                                             \
                    lds_read_b128<((J) + 2) * 16>(n0, pa);                              \
                    lds_read_b128<((J) + 3) * 16>(n1, pa);                              \
                    lds_wait<2>(x0, x1, x2, x3);                                        \
                } else {                                                                \
                    lds_wait<0>(x0, x1, x2, x3);                                        \
                }                                                                       \
                fma_row(acc[(J)], p0.y, x0);                                            \
                fma_row(acc[(J)], p0.w, x1);                                            \
                fma_row(acc[(J) + 1], p1.y, x2);                                        \
                fma_row(acc[(J) + 1], p1.w, x3);                                        \
                if (MORE) { p0 = n0; p1 = n1; }                                         \
            }
            PGCN_STRIP_STEP(0, true) PGCN_STRIP_STEP(2, true) PGCN_STRIP_STEP(4, true) PGCN_STRIP_STEP(6, true)
            PGCN_STRIP_STEP(8, true) PGCN_STRIP_STEP(10, true) PGCN_STRIP_STEP(12, true) PGCN_STRIP_STEP(14, false)
#undef PGCN_STRIP_STEP
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // nobody reads the buffers of record k any more
        pb = pbn;
    }
    const int fcol = fcol0 + sub * 4;
    if (fcol < f) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int rit = j * NG + group;
            *reinterpret_cast<float4 *>(partial + ((int64_t)wk.w + rit) * f + fcol) =
                make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
        }
    }
}

// Any width / alignment: same records, operands read straight from global memory, one feature per
// lane (blockIdx.y walks the features 32 at a time).  Correctness path, not a fast path.
__global__ __launch_bounds__(kThreads, 1) void spmm_strip_generic_kernel(
    const int4 *__restrict__ work, const int4 *__restrict__ recs, const int32_t *__restrict__ pairs,
    const float *__restrict__ B, int64_t ldb, int32_t f, float *__restrict__ partial) {
    const int4 wk = work[blockIdx.x];
    const int sub = threadIdx.x & 31;
    const int group = threadIdx.x >> 5;
    const int fcol = blockIdx.y * 32 + sub;
    const bool fact = fcol < f;
    float acc[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) acc[j] = 0.f;
    for (int k = wk.y; k < wk.z; ++k) {
        const int64_t col0 = (int64_t)recs[k].x * TC;
        const int32_t *pp = pairs + ((int64_t)k * TRS + group * RW) * SB * 2;
#pragma unroll
        for (int j = 0; j < RW; ++j) {
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int32_t off = pp[(j * SB + u) * 2];
                const float w = __int_as_float(pp[(j * SB + u) * 2 + 1]);
                if (off != kPadOff && fact) acc[j] = fmaf(w, B[(col0 + (off >> 9)) * ldb + fcol], acc[j]);
            }
        }
    }
    if (fact) {
#pragma unroll
        for (int j = 0; j < RW; ++j) partial[((int64_t)wk.w + j * NG + group) * f + fcol] = acc[j];
    }
}

}  // namespace

extern "C" int pgcn_spmm_strip_f32(const int32_t *work, int64_t nwork, const int32_t *recs, const int32_t *pairs,
                                   const float *B, int64_t ldb, int64_t ncols, int32_t f, float *partial_ws,
                                   int64_t partial_ws_elems, int64_t nslots_total, pgcn_stream_t stream) {
    if (nwork < 0 || f <= 0 || ldb < f || ncols <= 0) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_f32: bad sizes");
    if (nwork == 0) return PGCN_OK;
    if (!work || !recs || !pairs || !B || !partial_ws)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_f32: null pointer");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_strip_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_f32: work list too long");
    if ((uintptr_t)pairs % 16 || (uintptr_t)recs % 16 || (uintptr_t)work % 16)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_f32: work / recs / pairs must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int4 *w4 = reinterpret_cast<const int4 *>(work);
    const int4 *r4 = reinterpret_cast<const int4 *>(recs);
    const bool vec = f % 4 == 0 && ldb % 4 == 0 && (uintptr_t)B % 16 == 0 && (uintptr_t)partial_ws % 16 == 0;
    static const int probe = getenv("PGCN_STRIP_PROBE") ? atoi(getenv("PGCN_STRIP_PROBE")) : 0;   // measurement aid, see the kernel
    if (vec) {
        int dev = 0;
        PGCN_HIP_CHECK(hipGetDevice(&dev));
        static bool attr_set[64] = {false};
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {   // the attribute is per device
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        const dim3 grid((unsigned)nwork, (unsigned)((f + 127) / 128));
        switch (probe) {
            case 1: hipLaunchKernelGGL(spmm_strip_kernel<1>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws); break;
            case 2: hipLaunchKernelGGL(spmm_strip_kernel<2>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws); break;
            case 3: hipLaunchKernelGGL(spmm_strip_kernel<3>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws); break;
            default: hipLaunchKernelGGL(spmm_strip_kernel<0>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws); break;
        }
    } else {
        hipLaunchKernelGGL(spmm_strip_generic_kernel, dim3((unsigned)nwork, (unsigned)((f + 31) / 32)), dim3(kThreads), 0, s,
                           w4, r4, pairs, B, ldb, f, partial_ws);
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
