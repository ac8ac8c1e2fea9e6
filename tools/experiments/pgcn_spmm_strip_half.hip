// pgcn_spmm_strip_half.hip -- the strip kernel in a shape that SHARES a CU with the gather kernel.
//
// Why.  The gather part of the SpMM (pgcn_spmm.hip) is bound by memory latency, not by occupancy: with 12 instead of
// 24 waves per CU it runs at the same speed (r02 probe: unused dynamic LDS capping the resident workgroups, 871 us
// at 6, 4 and 3 workgroups per CU, 960 us at 2).  The strip kernel (pgcn_spmm_strip.hip) is bound by LDS bandwidth
// and takes a whole CU: 1 024 threads x 120 registers, 157 KB of LDS.  Run one after the other they idle each
// other's resource.  This variant works on the SAME records with half the footprint -- 64 features per workgroup
// (grid y walks the features), so 32 KB panels, 81 KB of LDS and 4 row slots x 8 features = 32 accumulator
// registers per lane of its 1 024 threads -- so that one strip workgroup (16 waves: the LDS pipe needs them) and
// two gather workgroups are resident together when the two kernels are launched on two streams
// (kernels.HipKernels, PGCN_CORE_OVERLAP): the LDS-bound and the latency-bound work overlap on every CU.  (A first
// cut with 512 threads per workgroup overlapped perfectly -- the gather kernel kept its 874 us -- but took 1.8 ms
// itself: 8 waves do not keep the LDS pipe busy across the barriers and copy waits.)
//
// Mapping.  A GROUP is 8 lanes and a lane owns 8 features: chunks c and c ^ 8 of the 16 float4 chunks of a staged
// 256-byte row, c = (lane & 7) ^ 8 for the groups 2, 3, 6, 7 of a wave -- the four quads of every ds_read_b128
// phase group ({0-3,12-15,20-27}, ...) then hit four different quarters of the 64 banks whatever rows they read
// (conflict-free), and the second chunk's address is the first one's XOR 128 (rows are 256-byte aligned).  One
// pair read serves EIGHT groups and is followed by sixteen row reads.  128 groups x 4 row slots on the record
// layout of pgcn_spmm_strip.hip (64 x 8, local row = j * 64 + g): kernel group G takes the row slots 4 (G & 1) ..
// 4 (G & 1) + 3 of layout group G >> 1 -- 64 contiguous bytes of pairs -- byte offsets halved on the fly (512 ->
// 256 B rows).  Pipeline as there: wave-private pair ring (512 B per wave and record + the next record's 16-byte
// header, one record ahead), one barrier per run of records sharing a panel, the next run's panel copied between the compute
// steps of the run's first record, LDS reads in inline asm with hand-counted lgkmcnt.  Bit-identical results.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pgcn_internal.h"

namespace {

constexpr int TRS = PGCN_STRIP_TR;     // 512 rows per strip tile
constexpr int TC = PGCN_CORE_TC;       // 128 columns per panel
constexpr int SB = PGCN_STRIP_B;       // pair slots per row and record
constexpr int FW = 64;                 // features per workgroup
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int NG = kThreads / 8;       // 128 groups of 8 lanes
constexpr int RW = TRS / NG;           // 4 row slots per group
constexpr int LNG = 64, LRW = 8;       // the record layout: 64 groups x 8 row slots, local row = j * 64 + g
static_assert(RW == 4 && NG == 128 && SB == 2 && LNG * LRW == TRS, "layout constants are baked into the record format");

constexpr int kRowBytes = FW * 4;                     // 256 B of a staged row
constexpr int kPanelBytes = (TC + 1) * kRowBytes;     // 128 rows + the all-zero row (pair offset 65536 >> 1)
constexpr int kRecBytes = TRS * SB * 8;               // 8 KB of pairs per record
constexpr int kWaveRec = kRecBytes / kWaves;          // 512 B: the pairs of one wave's eight groups
constexpr int kSlotBytes = kWaveRec + 16;             // ... followed by the next record's 16-byte header
constexpr int kRing = 2;                              // ring slots per wave (record k, k + 1)
constexpr int kOffRing = 2 * kPanelBytes;
constexpr int kSmem = kOffRing + kWaves * kRing * kSlotBytes;
static_assert(kSmem > 80 * 1024 && kSmem < 96 * 1024, "one strip workgroup per CU, room for the gather kernel's LDS");

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// wait until at most n of this wave's asynchronous copies are outstanding (n is wave-uniform)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (__builtin_amdgcn_readfirstlane(n)) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    }
}

// LDS reads the compiler does not see as memory operations (a visible ds_read makes it wait for ALL outstanding
// asynchronous copies); the matching waits take the results as read-write operands.
template <int OFF>
__device__ __forceinline__ void lds_read_b128(f32x4 &v, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_wait0(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d, f32x4 &e) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}
__device__ __forceinline__ void lds_wait1(f32x4 &a) {
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a));
}

__device__ __forceinline__ int64_t panel_base(int panel, int64_t ncols) {
    const int64_t col0 = (int64_t)panel * TC;        // the LAST panel of an operand with ncols % 128 != 0 is the window
    return col0 + TC <= ncols ? col0 : ncols - TC;   // [ncols - 128, ncols): no copy reads past the operand
}

// Asynchronous copies Q0..Q1-1 (of 2 per thread) of a panel: 128 rows x 256 B, 16 B per lane, LDS image lane-linear.
// Source = (wave-uniform 64-bit base of the quarter panel) + (one 32-bit per-lane offset, the same for every panel):
// the SGPR-base form of global_load_lds, so a copy issued between the compute steps needs no address registers.
template <int Q0, int Q1>
__device__ __forceinline__ void issue_panel(int panel, uint32_t lds0, int pb, const float *__restrict__ B, int64_t ldb,
                                            int64_t ncols, int fcol0, uint32_t lane_off, int wave) {
    const int64_t col0 = panel_base(panel, ncols);
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        const char *base = reinterpret_cast<const char *>(B + (col0 + q * 64) * ldb + fcol0);   // wave-uniform
        const uint32_t dst = lds0 + pb * kPanelBytes + (q * kThreads + wave * 64) * 16;         // wave-uniform
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(lane_off), "s"(base) : "memory", "m0");
    }
}

// One of the two copies, skipped when `on` (wave-uniform) is 0.  The branch lives INSIDE the asm statement: a C++
// branch between the compute steps splits the straight-line block of hand-scheduled LDS reads and the register
// allocator gives up (100+ spills).
template <int Q>
__device__ __forceinline__ void issue_panel_if(int on, int panel, uint32_t lds0, int pb, const float *__restrict__ B, int64_t ldb,
                                               int64_t ncols, int fcol0, uint32_t lane_off, int wave) {
    const int64_t col0 = panel_base(panel, ncols);
    const char *base = reinterpret_cast<const char *>(B + (col0 + Q * 64) * ldb + fcol0);
    const uint32_t dst = lds0 + pb * kPanelBytes + (Q * kThreads + wave * 64) * 16;
    asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n1:"
                 :: "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(lane_off), "s"(base), "s"(__builtin_amdgcn_readfirstlane(on)) : "memory", "m0", "scc");
}

// One asynchronous copy per wave into ring slot `slot`: lanes 0-31 the 512 B of pairs of this wave's eight groups of
// record k, lane 32 the 16-byte header of record k + 1 behind them -- no scalar loads in the loop (an outstanding
// s_load would sit in lgkmcnt behind every counted LDS wait).
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__device__ __forceinline__ void issue_record(const int32_t *__restrict__ pairs, const int4 *__restrict__ recs, int64_t k, int64_t k1,
                                             int slot, char *smem, int wave, int lane) {
    if (lane <= 32) {
        const int32_t *src = lane < 32 ? pairs + k * (int64_t)(TRS * SB * 2) + wave * (kWaveRec / 4) + lane * 4
                                       : reinterpret_cast<const int32_t *>(recs + (k + 1 < k1 ? k + 1 : k));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + kOffRing + (wave * kRing + slot) * kSlotBytes), 16, 0, 0);
    }
}

__device__ __forceinline__ void fma_chunk(f32x2 (&acc)[2], float w, const f32x4 &x) {
    const f32x2 ww = {w, w};
    acc[0] = __builtin_elementwise_fma(ww, f32x2{x.x, x.y}, acc[0]);
    acc[1] = __builtin_elementwise_fma(ww, f32x2{x.z, x.w}, acc[1]);
}

// One record (layer) of the piece: 4 row slots x 2 entries for this lane's group.  ISSUE (wave-uniform): the two
// copies of the next run's panel go out between the steps.  `hn` receives the header of the NEXT record (kept behind
// this record's pairs in the ring slot), read under the last steps.
__device__ __forceinline__ void compute_record(const int ISSUE, f32x2 (&acc)[RW][2][2], const uint32_t pa, const uint32_t rowbase, f32x4 &hn,
                                               uint32_t hdr_lds, int next_panel, uint32_t lds0, int pbn, const float *__restrict__ B, int64_t ldb,
                                               int64_t ncols, int fcol0, uint32_t lane_off, int wave) {
    f32x4 pA, pB, pC, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3;
    lds_read_b128<0>(pA, pa);
    lds_read_b128<16>(pB, pa);
    lds_wait1(pA);
    {
        const uint32_t a0 = rowbase + ((uint32_t)__float_as_int(pA.x) >> 1), a1 = rowbase + ((uint32_t)__float_as_int(pA.z) >> 1);
        lds_read_b128<0>(xa0, a0);
        lds_read_b128<0>(xa1, a0 ^ 128u);
        lds_read_b128<0>(xa2, a1);
        lds_read_b128<0>(xa3, a1 ^ 128u);
    }
    // step J (P = pairs of slot J, N = pairs of slot J + 1, M = free): the rows of slot J and N have landed; the rows
    // of slot J + 1 and the pairs of slot J + 2 go out BEFORE the packed FMAs of slot J (sched_barrier pins that
    // order), so the LDS pipe always holds five reads of this wave
#define PGCN_STRIP_STEP(J, P, N, M, XA0, XA1, XA2, XA3, XB0, XB1, XB2, XB3)               \
    {                                                                                      \
        if ((J) + 1 < RW) {                                                                \
            lds_wait0(XA0, XA1, XA2, XA3, N);                                              \
            const uint32_t a0 = rowbase + ((uint32_t)__float_as_int(N.x) >> 1);            \
            const uint32_t a1 = rowbase + ((uint32_t)__float_as_int(N.z) >> 1);            \
            lds_read_b128<0>(XB0, a0);                                                     \
            lds_read_b128<0>(XB1, a0 ^ 128u);                                              \
            lds_read_b128<0>(XB2, a1);                                                     \
            lds_read_b128<0>(XB3, a1 ^ 128u);                                              \
            if ((J) + 2 < RW) { lds_read_b128<(((J) + 2) % RW) * 16>(M, pa); }             \
            else { lds_read_b128<0>(hn, hdr_lds); }                                        \
        } else {                                                                           \
            lds_wait0(XA0, XA1, XA2, XA3, hn);                                             \
        }                                                                                  \
        if (((J) & 1) == 0)                                                                \
            issue_panel_if<(J) / 2>(ISSUE, next_panel, lds0, pbn, B, ldb, ncols, fcol0, lane_off, wave); \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        fma_chunk(acc[(J)][0], P.y, XA0);                                                  \
        fma_chunk(acc[(J)][1], P.y, XA1);                                                  \
        fma_chunk(acc[(J)][0], P.w, XA2);                                                  \
        fma_chunk(acc[(J)][1], P.w, XA3);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
    PGCN_STRIP_STEP(0, pA, pB, pC, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3)
    PGCN_STRIP_STEP(1, pB, pC, pA, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3)
    PGCN_STRIP_STEP(2, pC, pA, pB, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3)
    PGCN_STRIP_STEP(3, pA, pB, pC, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3)
#undef PGCN_STRIP_STEP
}

// work / recs / pairs: as pgcn_spmm_strip_f32 (same records).  PROBE (PGCN_STRIP_PROBE): 1 = no compute phase,
// 2 = no panel staging.
template <int PROBE>
__global__ __launch_bounds__(kThreads, 1) void spmm_strip_co_kernel(
    const int4 *__restrict__ work, const int4 *__restrict__ recs, const int32_t *__restrict__ pairs,
    const float *__restrict__ B, int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    int4 wk = work[blockIdx.x];
    wk.x = __builtin_amdgcn_readfirstlane(wk.x); wk.y = __builtin_amdgcn_readfirstlane(wk.y);   // wave-uniform: scalar control flow
    wk.z = __builtin_amdgcn_readfirstlane(wk.z); wk.w = __builtin_amdgcn_readfirstlane(wk.w);
    const int fcol0 = blockIdx.y * FW;
    const int fw = min(FW, f - fcol0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gq = lane >> 3;              // group inside the wave
    const int G = wave * 8 + gq;           // 0..127: row slots 4 (G & 1) .. + 3 of layout group G >> 1
    const int c0 = (lane & 7) ^ ((gq & 2) << 2);   // this lane's chunks of a row: c0 and c0 ^ 8
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    // panel copies: thread t moves 16 B of row (q * 64 + t / 16), chunk t % 16 (narrow panels repeat their last chunk)
    const int c4s = min((int)(threadIdx.x & 15), (fw >> 2) - 1);
    const uint32_t lane_off = (uint32_t)(((int64_t)(threadIdx.x >> 4) * ldb + c4s * 4) * 4);

    f32x2 acc[RW][2][2];
#pragma unroll
    for (int j = 0; j < RW; ++j) acc[j][0][0] = acc[j][0][1] = acc[j][1][0] = acc[j][1][1] = f32x2{0.f, 0.f};
    if (threadIdx.x < 32)   // the all-zero row of both panel buffers
        *reinterpret_cast<float4 *>(smem + (threadIdx.x >> 4) * kPanelBytes + TC * kRowBytes + (threadIdx.x & 15) * 16) =
            make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... has left this wave before the first barrier

    const int k0 = wk.y, k1 = wk.z;
    int pb = 0, pbn = 0;
    int flags, next_panel;                 // header of the current record; the later ones arrive behind the pairs
    {
        const int4 rc = recs[k0];
        flags = __builtin_amdgcn_readfirstlane(rc.y); next_panel = __builtin_amdgcn_readfirstlane(rc.z);
        if (!(PROBE & 2)) issue_panel<0, 2>(__builtin_amdgcn_readfirstlane(rc.x), lds0, 0, B, ldb, ncols, fcol0, lane_off, wave);
    }
    issue_record(pairs, recs, k0, k1, 0, smem, wave, lane);
    // vmcnt bookkeeping (copies complete in issue order): what this wave issued AFTER the copies it waits for
    int after_panel = 1;                   // ... after the latest panel copies (saturates at 1)
    bool issued1 = false;                  // panel copies went out during the previous record
    int slot = 0, in_run = 0;
    for (int k = k0; k < k1; ++k) {
        // issue order of an iteration: [pairs k + 1 with header k + 2] [panel of the next run, during the compute phase]
        wait_vm_dyn(issued1 ? 2 : 0);      // pairs of record k (and the header behind them) have landed
        if (k + 1 < k1) {
            issue_record(pairs, recs, k + 1, k1, slot ^ 1, smem, wave, lane);
            after_panel = 1;
        }
        // The issue arbiter serves the waves of a SIMD oldest first: a wave that is ahead lowers its own priority
        // (3, 2, 1, 0 for the first, second, ... record after a barrier) so that the waves of a run finish together.
        if (!(flags & 1)) in_run = 0;
        switch (in_run) {
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
        ++in_run;
        bool cur_issued = false;
        if (!(flags & 1)) {                // this record starts a run of a new panel
            wait_vm_dyn(after_panel);      // this wave's copies of the panel have landed
            __builtin_amdgcn_s_barrier();  // ... everybody's have, and nobody reads the other buffer any more
            pb = pbn;
            if (next_panel >= 0 && !(PROBE & 2)) {
                pbn = pb ^ 1;
                cur_issued = true;         // (the copies go out inside compute_record)
                after_panel = 0;
            }
        }
        const uint32_t slot_lds = lds0 + kOffRing + (wave * kRing + slot) * kSlotBytes;
        const uint32_t pa = slot_lds + gq * (RW * SB * 8);
        const uint32_t rowbase = lds0 + pb * kPanelBytes + c0 * 16;
        f32x4 hn;
        if constexpr (!(PROBE & 1)) {
            compute_record(cur_issued ? 1 : 0, acc, pa, rowbase, hn, slot_lds + kWaveRec, next_panel, lds0, pbn, B, ldb, ncols, fcol0,
                           lane_off, wave);
        } else {
            if (cur_issued) issue_panel<0, 2>(next_panel, lds0, pbn, B, ldb, ncols, fcol0, lane_off, wave);
            lds_read_b128<0>(hn, slot_lds + kWaveRec);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hn));
        }
        flags = __builtin_amdgcn_readfirstlane(__float_as_int(hn.y));
        next_panel = __builtin_amdgcn_readfirstlane(__float_as_int(hn.z));
        issued1 = cur_issued;
        slot ^= 1;
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        float *o = partial + ((int64_t)wk.w + ((G & 1) * RW + j) * LNG + (G >> 1)) * f + fcol0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int fc = (c0 ^ (c * 8)) * 4;
            if (fc < fw)
                *reinterpret_cast<float4 *>(o + fc) = make_float4(acc[j][c][0].x, acc[j][c][0].y, acc[j][c][1].x, acc[j][c][1].y);
        }
    }
}

}  // namespace

extern "C" int pgcn_spmm_strip_half_f32(const int32_t *work, int64_t nwork, const int32_t *recs, const int32_t *pairs,
                                        const float *B, int64_t ldb, int64_t ncols, int32_t f, float *partial_ws,
                                        int64_t partial_ws_elems, int64_t nslots_total, pgcn_stream_t stream) {
    if (nwork < 0 || f <= 0 || ldb < f || ncols < TC) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_half_f32: bad sizes (a panel is 128 rows of B)");
    if (nwork == 0) return PGCN_OK;
    if (!work || !recs || !pairs || !B || !partial_ws)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_half_f32: null pointer");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_strip_half_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_half_f32: work list too long");
    if ((uintptr_t)pairs % 16 || (uintptr_t)recs % 16 || (uintptr_t)work % 16)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_half_f32: work / recs / pairs must be 16-byte aligned");
    if (!(f % 4 == 0 && ldb % 4 == 0 && (uintptr_t)B % 16 == 0 && (uintptr_t)partial_ws % 16 == 0 && ldb < (1 << 24)))
        return pgcn_set_error(PGCN_EUNSUPPORTED, "pgcn_spmm_strip_half_f32: needs f % 4 == 0 and 16-byte aligned operands (use pgcn_spmm_strip_f32)");
    hipStream_t s = (hipStream_t)stream;
    const int4 *w4 = reinterpret_cast<const int4 *>(work);
    const int4 *r4 = reinterpret_cast<const int4 *>(recs);
    static const int probe = getenv("PGCN_STRIP_PROBE") ? atoi(getenv("PGCN_STRIP_PROBE")) : 0;   // measurement aid, see the kernel
    int dev = 0;
    PGCN_HIP_CHECK(hipGetDevice(&dev));
    static bool attr_set[64] = {false};
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {   // the attribute is per device
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_co_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_co_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_co_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const dim3 grid((unsigned)nwork, (unsigned)((f + FW - 1) / FW));
    switch (probe) {
        case 1: hipLaunchKernelGGL(spmm_strip_co_kernel<1>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws); break;
        case 2: hipLaunchKernelGGL(spmm_strip_co_kernel<2>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws); break;
        default: hipLaunchKernelGGL(spmm_strip_co_kernel<0>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws); break;
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
