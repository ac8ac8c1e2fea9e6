// pgcn_spmm_strip.hip -- LDS-staged SpMM over TALL tiles (512 rows x 128 columns) for gfx950.
//
// Why.  A gather through the vector L1 moves one 512 B row of the dense operand per stored entry and saturates
// near 18 TB/s on an MI355X even when every row hits in L2 (r02 probe, uniform hot set): at 58.8 GB of gathers per
// Reddit-sized SpMM that is the whole budget.  Staging a 128-row panel of B once in LDS (64 KB through the same L1
// path) and serving all entries of a 512-row tile from LDS replaces 512 B per ENTRY by 64 KB per TILE (part of
// /root/reference/GPU/PGCN.py:127's torch.sparse.mm).
//
// Records.  A RECORD is one LAYER of a 512 x 128 tile: the (2 l)-th and (2 l + 1)-th stored entry of every row,
// i.e. exactly 2 {byte offset of the column's row in the staged panel, value} pairs per row = 8 KB; an unused slot
// points at an all-zero LDS row with value 0.0 (no row the matrix does not reference is ever combined: no 0 x Inf).
// With a fixed shape the compute phase is straight-line code: no counts, no branches.
//
// Mapping (second generation, r02; tools/micro/lds_rate.hip and tools/micro/strip_bench.cpp are the measurements):
//   * the LDS serves whole-row reads at 232 B/clk/CU, but a ds_read_b128 that BROADCASTS 16 B of pairs to a group
//     costs as much as one that reads 1 KB of rows.  A GROUP is 16 lanes and a lane owns 8 features (chunks s and
//     s + 16 of the 32 float4 chunks of a row), so one pair read serves FOUR groups (4 x 2 entries) and is followed
//     by eight row reads: pairs take 1/5 of the LDS cycles (1/3 with 32-lane groups).  Lanes of a ds_read_b128
//     phase group ({0-3,12-15,20-27}, ...) read chunk (lane mod 16) of their rows -- the bank pattern of a linear
//     read: conflict-free for any mix of rows.  64 groups x 8 row slots: local row = j * 64 + g, 8 x 2 float4
//     accumulators per lane over the whole piece.
//   * pairs are wave-private: wave w copies the 512 B of its four groups (global_load_lds, lanes 0-31; lane 32
//     brings the next record's 16-byte header) into its own three-slot ring two records ahead and waits for them
//     with a counted vmcnt -- no barrier, no scalar loads in the loop (an outstanding s_load sits in lgkmcnt behind
//     every counted LDS wait).
//   * a workgroup barrier is only needed when the PANEL changes (every ~3 records on the benchmark graph): one per
//     run of records that share a panel.  The next run's panel is copied into the other buffer during the current
//     run; its four copies per thread go out BETWEEN the compute steps of the run's first record (SGPR-base form,
//     no address registers), not in a burst behind the barrier.
//   * LDS reads are inline asm with hand-counted lgkmcnt (a ds_read the compiler can see makes it wait for ALL
//     outstanding asynchronous copies); sched_barrier pins "next slot's reads before this slot's FMAs".
// Deterministic: fixed order, no atomics; a piece writes 512 partial rows to slots that pgcn_spmm_fixup_f32 adds
// in list order.  Bit-identical to the first-generation kernel on the same records (strip_bench.cpp), 1.27x faster.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pgcn_internal.h"
#include "pgcn_once.h"

#pragma clang diagnostic ignored "-Winline-asm"   // the copies set M0 themselves ("m0" on the clobber list)

namespace {

constexpr int TRS = PGCN_STRIP_TR;     // 512 rows per strip tile
constexpr int TC = PGCN_CORE_TC;       // 128 columns per panel
constexpr int SB = PGCN_STRIP_B;       // pair slots per row and record
constexpr int kThreads = 1024;
constexpr int NG = kThreads / 16;      // 64 groups of 16 lanes
constexpr int RW = TRS / NG;           // 8 row slots per group: local row = j * 64 + g
static_assert(RW == 8 && NG == 64 && SB == 2, "layout constants are baked into the record format");

constexpr int kPanelBytes = (TC + 1) * 512;           // 128 rows x 128 fp32 + the all-zero row
constexpr int kPadOff = TC * 512;                     // what an unused pair slot points at
constexpr int kRecBytes = TRS * SB * 8;               // 8 KB of pairs per record
constexpr int kWaveRec = kRecBytes / (kThreads / 64); // 512 B: the pairs of one wave's four groups
constexpr int kRing = 3;                              // ring slots per wave (record k, k + 1, k + 2)
constexpr int kSlotBytes = kWaveRec + 16;             // ... each followed by the record's 16-byte header
constexpr int kOffRing = 2 * kPanelBytes;
constexpr int kSmem = kOffRing + (kThreads / 64) * kRing * kSlotBytes;
static_assert(kSmem <= 160 * 1024, "LDS budget of one CU");

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

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
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
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
__device__ __forceinline__ void lds_wait0_4(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_wait1(f32x4 &a) {
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a));
}

// 4 asynchronous copies per thread: the 128 rows of a panel, 16 B per lane, LDS image lane-linear
// Asynchronous copies Q0..Q1-1 (of 4 per thread) of a panel: 128 rows x 512 B, 16 B per lane, LDS image lane-linear.
// Source = (wave-uniform 64-bit base of the quarter panel) + (one 32-bit per-lane offset, the same for every panel):
// the SGPR-base form of global_load_lds, so a copy issued between the compute steps needs no address registers.
// The LAST panel of an operand whose row count is not a multiple of 128 is the window [ncols - 128, ncols): the
// host stores its pair offsets relative to that base, no copy ever reads past the operand.
template <int Q0, int Q1>
__device__ __forceinline__ void issue_panel(int panel, uint32_t lds0, int pb, const float *__restrict__ B, int64_t ldb,
                                            int64_t ncols, int fcol0, uint32_t lane_off, int wave) {
    int64_t col0 = (int64_t)panel * TC;
    col0 = col0 + TC <= ncols ? col0 : ncols - TC;
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        const char *base = reinterpret_cast<const char *>(B + (col0 + q * 32) * ldb + fcol0);   // wave-uniform
        const uint32_t dst = lds0 + pb * kPanelBytes + (q * kThreads + wave * 64) * 16;         // wave-uniform
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(dst), "v"(lane_off), "s"(base) : "memory", "m0");
    }
}

// One of the four copies, skipped when `on` (wave-uniform) is 0.  The branch lives INSIDE the asm statement: a C++
// branch between the compute steps splits the straight-line block of 40 hand-scheduled LDS reads and the register
// allocator gives up (100+ spills).
template <int Q>
__device__ __forceinline__ void issue_panel_if(int on, int panel, uint32_t lds0, int pb, const float *__restrict__ B, int64_t ldb,
                                               int64_t ncols, int fcol0, uint32_t lane_off, int wave) {
    int64_t col0 = (int64_t)panel * TC;
    col0 = col0 + TC <= ncols ? col0 : ncols - TC;
    const char *base = reinterpret_cast<const char *>(B + (col0 + Q * 32) * ldb + fcol0);
    const uint32_t dst = lds0 + pb * kPanelBytes + (Q * kThreads + wave * 64) * 16;
    asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n1:"
                 :: "s"(__builtin_amdgcn_readfirstlane(dst)), "v"(lane_off), "s"(base), "s"(__builtin_amdgcn_readfirstlane(on)) : "memory", "m0", "scc");
}

// 1 asynchronous copy per wave into ring slot `slot`: lanes 0-31 the pairs of this wave's four groups of record k,
// lane 32 the header of record k + 1 (so the loop needs no scalar loads: an outstanding s_load would sit in lgkmcnt
// behind every counted LDS wait, and a header line missing the scalar cache stalls all 16 waves)
__device__ __forceinline__ void issue_pairs(const int32_t *__restrict__ pairs, const int4 *__restrict__ recs, int64_t k, int64_t k1,
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

// One record (layer) of the piece: 8 row slots x 2 entries for this lane's group.  ISSUE: the four copies of the
// next run's panel go out between the steps (all at once they keep every wave of the workgroup stuck in the issue
// queue for ~2 k clk right after the barrier).  `hn` receives the header of the NEXT record (kept behind this
// record's pairs in the ring slot), read under the last steps.
__device__ __forceinline__ void compute_record(const int ISSUE, f32x2 (&acc)[RW][2][2], const uint32_t pa, const uint32_t rowbase, f32x4 &hn,
                                               uint32_t gq128, int next_panel, uint32_t lds0, int pbn, const float *__restrict__ B, int64_t ldb,
                                               int64_t ncols, int fcol0, uint32_t lane_off, int wave) {
    f32x4 pA, pB, pC, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3;
    lds_read_b128<0>(pA, pa);
    lds_read_b128<16>(pB, pa);
    lds_wait1(pA);
    {
        const uint32_t a0 = rowbase + (uint32_t)__float_as_int(pA.x), a1 = rowbase + (uint32_t)__float_as_int(pA.z);
        lds_read_b128<0>(xa0, a0);
        lds_read_b128<256>(xa1, a0);
        lds_read_b128<0>(xa2, a1);
        lds_read_b128<256>(xa3, a1);
    }
    // step J (P = pairs of slot J, N = pairs of slot J + 1, M = free): the rows of slot J and N have landed; the rows
    // of slot J + 1 and the pairs of slot J + 2 go out BEFORE the packed FMAs of slot J (sched_barrier pins that
    // order), so the LDS pipe always holds five reads of this wave
#define PGCN_STRIP_STEP(J, P, N, M, XA0, XA1, XA2, XA3, XB0, XB1, XB2, XB3)              \
    {                                                                                      \
        if ((J) + 1 < RW) {                                                                \
            lds_wait0(XA0, XA1, XA2, XA3, N);                                              \
            const uint32_t a0 = rowbase + (uint32_t)__float_as_int(N.x);                   \
            const uint32_t a1 = rowbase + (uint32_t)__float_as_int(N.z);                   \
            lds_read_b128<0>(XB0, a0);                                                     \
            lds_read_b128<256>(XB1, a0);                                                   \
            lds_read_b128<0>(XB2, a1);                                                     \
            lds_read_b128<256>(XB3, a1);                                                   \
            if ((J) + 2 < RW) { lds_read_b128<(((J) + 2) % RW) * 16>(M, pa); }             \
            else { lds_read_b128<kWaveRec>(hn, pa - gq128); }                               \
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
    PGCN_STRIP_STEP(4, pB, pC, pA, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3)
    PGCN_STRIP_STEP(5, pC, pA, pB, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3)
    PGCN_STRIP_STEP(6, pA, pB, pC, xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3)
    PGCN_STRIP_STEP(7, pB, pC, pA, xb0, xb1, xb2, xb3, xa0, xa1, xa2, xa3)
#undef PGCN_STRIP_STEP
}

// work: int4 {tile row, first record, one-past-last record, first slot}
// recs: int4 {panel, flags, next panel, layer}; flags bit 0: the panel is the one of the previous record of the
//       piece; `next panel` (records that start a run of one panel): panel of the piece's NEXT run or -1
// PROBE (measurement aid, PGCN_STRIP_PROBE): 1 = no compute phase, 2 = no panel staging, 4 = phase timers
template <int PROBE>
__global__ __launch_bounds__(kThreads, 1) void spmm_strip_kernel(
    const int4 *__restrict__ work, const int4 *__restrict__ recs, const int32_t *__restrict__ pairs,
    const float *__restrict__ B, int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4 wk = work[blockIdx.x];
    wk.x = __builtin_amdgcn_readfirstlane(wk.x); wk.y = __builtin_amdgcn_readfirstlane(wk.y);   // wave-uniform: scalar control flow
    wk.z = __builtin_amdgcn_readfirstlane(wk.z); wk.w = __builtin_amdgcn_readfirstlane(wk.w);
    const int fcol0 = blockIdx.y * 128;
    const int fw = min(128, f - fcol0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane & 15;
    const int gq = (lane >> 4);            // group inside the wave
    const int g = wave * 4 + gq;           // 0..63
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    // panel copies: thread t moves 16 B of row (q * 32 + t / 32), chunk t % 32 (narrow panels repeat their last chunk)
    const int c4s = min((int)(threadIdx.x & 31), (fw >> 2) - 1);
    const uint32_t lane_off = (uint32_t)(((int64_t)(threadIdx.x >> 5) * ldb + c4s * 4) * 4);

    f32x2 acc[RW][2][2];
#pragma unroll
    for (int j = 0; j < RW; ++j) acc[j][0][0] = acc[j][0][1] = acc[j][1][0] = acc[j][1][1] = f32x2{0.f, 0.f};
    if (threadIdx.x < 64)   // the all-zero row of both panel buffers
        *reinterpret_cast<float4 *>(smem + (threadIdx.x >> 5) * kPanelBytes + kPadOff + (threadIdx.x & 31) * 16) =
            make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... has left this wave before the first barrier

    const int k0 = wk.y, k1 = wk.z;
    int4 rc = recs[k0];                    // header of the current record; the later ones arrive with the pairs
    rc.x = __builtin_amdgcn_readfirstlane(rc.x); rc.y = __builtin_amdgcn_readfirstlane(rc.y);
    rc.z = __builtin_amdgcn_readfirstlane(rc.z);
    int pb = 0, pbn = 0;
    if (!(PROBE & 2)) issue_panel<0, 4>(rc.x, lds0, 0, B, ldb, ncols, fcol0, lane_off, wave);
    issue_pairs(pairs, recs, k0, k1, 0, smem, wave, lane);
    bool have_next = k0 + 1 < k1;          // the pairs of record k + 1 are in flight
    if (have_next) issue_pairs(pairs, recs, k0 + 1, k1, 1, smem, wave, lane);
    // vmcnt bookkeeping (copies complete in issue order): what this wave issued AFTER the copies it waits for
    int after_panel = have_next ? 2 : 1;   // ... after the latest panel copies (saturates at 2)
    bool issued1 = false, issued2 = false; // panel copies went out during the previous record / the one before
    int slot = 0, in_run = 0;
    long long t_vm = 0, t_bar = 0, t_cmp = 0, t0 = 0, t1;
    for (int k = k0; k < k1; ++k) {
        if constexpr (PROBE & 4) t0 = __builtin_readcyclecounter();
        // issue order of an iteration: [pairs k + 2] [panel of the next run, during the compute phase]
        int nafter = (issued2 ? 4 : 0) + (issued1 ? 4 : 0) + (k + 1 < k1 ? 1 : 0);
        if (k + 2 < k1) {
            int s2 = slot + 2; s2 = s2 >= kRing ? s2 - kRing : s2;
            issue_pairs(pairs, recs, k + 2, k1, s2, smem, wave, lane);
            after_panel = after_panel < 2 ? after_panel + 1 : 2;
            ++nafter;
        }
        wait_vm_dyn(nafter);               // the pairs of record k have landed in this wave's ring
        if constexpr (PROBE & 4) { t1 = __builtin_readcyclecounter(); t_vm += t1 - t0; t0 = t1; }
        // The issue arbiter serves the waves of a SIMD oldest first: without help wave 0 finishes a run 1.5x sooner
        // than wave 15 and idles at the next barrier while the LDS pipe starves on the few waves left.  A wave that
        // is ahead lowers its own priority (3, 2, 1, 0 for the first, second, ... record after a barrier).
        if (!(rc.y & 1)) in_run = 0;
        switch (in_run) {
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
        ++in_run;
        bool cur_issued = false;
        if (!(rc.y & 1)) {                 // this record starts a run of a new panel
            wait_vm_dyn(after_panel);      // this wave's copies of the panel have landed
            __builtin_amdgcn_s_barrier();  // ... everybody's have, and nobody reads the other buffer any more
            pb = pbn;
            if (rc.z >= 0 && !(PROBE & 2)) {
                pbn = pb ^ 1;
                cur_issued = true;         // (the copies go out inside compute_record)
                after_panel = 0;
            }
        }
        if constexpr (PROBE & 4) { t1 = __builtin_readcyclecounter(); t_bar += t1 - t0; t0 = t1; }
        const uint32_t pa = lds0 + kOffRing + (wave * kRing + slot) * kSlotBytes + gq * (RW * SB * 8);
        const uint32_t rowbase = lds0 + pb * kPanelBytes + s * 16;
        f32x4 hn;
        if constexpr (!(PROBE & 1)) {
            compute_record(cur_issued ? 1 : 0, acc, pa, rowbase, hn, gq * (RW * SB * 8), rc.z, lds0, pbn, B, ldb, ncols, fcol0, lane_off, wave);
        } else {
            if (cur_issued) issue_panel<0, 4>(rc.z, lds0, pbn, B, ldb, ncols, fcol0, lane_off, wave);
            lds_read_b128<kWaveRec>(hn, pa - gq * (RW * SB * 8));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hn));
        }
        rc.x = __builtin_amdgcn_readfirstlane(__float_as_int(hn.x));
        rc.y = __builtin_amdgcn_readfirstlane(__float_as_int(hn.y));
        rc.z = __builtin_amdgcn_readfirstlane(__float_as_int(hn.z));
        if constexpr (PROBE & 4) { t1 = __builtin_readcyclecounter(); t_cmp += t1 - t0; }
        issued2 = issued1;
        issued1 = cur_issued;
        slot = slot + 1 >= kRing ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PROBE & 4) {             // timer ticks per record of the phases, one row of the slot block per wave
        if (lane == 0) {
            float *o = partial + ((int64_t)wk.w + wave) * f + fcol0;
            const float nr = (float)(k1 - k0);
            o[0] = (float)t_vm / nr; o[1] = 0.f; o[2] = (float)t_bar / nr; o[3] = (float)t_cmp / nr;
        }
        return;
    }
    if (fw >= 128) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            float *o = partial + ((int64_t)wk.w + j * NG + g) * f + fcol0 + s * 4;
            *reinterpret_cast<float4 *>(o) = make_float4(acc[j][0][0].x, acc[j][0][0].y, acc[j][0][1].x, acc[j][0][1].y);
            *reinterpret_cast<float4 *>(o + 64) = make_float4(acc[j][1][0].x, acc[j][1][0].y, acc[j][1][1].x, acc[j][1][1].y);
        }
    } else {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            float *o = partial + ((int64_t)wk.w + j * NG + g) * f + fcol0 + s * 4;
            if (s * 4 < fw) *reinterpret_cast<float4 *>(o) = make_float4(acc[j][0][0].x, acc[j][0][0].y, acc[j][0][1].x, acc[j][0][1].y);
            if (64 + s * 4 < fw)
                *reinterpret_cast<float4 *>(o + 64) = make_float4(acc[j][1][0].x, acc[j][1][0].y, acc[j][1][1].x, acc[j][1][1].y);
        }
    }
}

// Any width / alignment: same records, operands read straight from global memory, one feature per
// lane (blockIdx.y walks the features 32 at a time).  Correctness path, not a fast path.
__global__ __launch_bounds__(kThreads, 1) void spmm_strip_generic_kernel(
    const int4 *__restrict__ work, const int4 *__restrict__ recs, const int32_t *__restrict__ pairs,
    const float *__restrict__ B, int64_t ldb, int64_t ncols, int32_t f, float *__restrict__ partial) {
    const int4 wk = work[blockIdx.x];
    const int sub = threadIdx.x & 31;
    const int g32 = threadIdx.x >> 5;      // 32 groups of 32 lanes, 16 rows each: row = i * 32 + g32
    const int fcol = blockIdx.y * 32 + sub;
    const bool fact = fcol < f;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = wk.y; k < wk.z; ++k) {
        int64_t col0 = (int64_t)recs[k].x * TC;
        col0 = col0 + TC <= ncols ? col0 : ncols - TC;   // the last panel is the window [ncols - 128, ncols)
        const int32_t *pr = pairs + (int64_t)k * TRS * SB * 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 32 + g32;
            const int32_t *pp = pr + (((row % NG) * RW + row / NG) * SB) * 2;
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int32_t off = pp[u * 2];
                const float w = __int_as_float(pp[u * 2 + 1]);
                if (off != kPadOff && fact) acc[i] = fmaf(w, B[(col0 + (off >> 9)) * ldb + fcol], acc[i]);
            }
        }
    }
    if (fact) {
#pragma unroll
        for (int i = 0; i < 16; ++i) partial[((int64_t)wk.w + i * 32 + g32) * f + fcol] = acc[i];
    }
}

}  // namespace

extern "C" int pgcn_spmm_strip_f32(const int32_t *work, int64_t nwork, const int32_t *recs, const int32_t *pairs,
                                    const float *B, int64_t ldb, int64_t ncols, int32_t f, float *partial_ws,
                                    int64_t partial_ws_elems, int64_t nslots_total, pgcn_stream_t stream) {
    if (nwork < 0 || f <= 0 || ldb < f || ncols < TC) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_strip_f32: bad sizes (a panel is 128 rows of B)");
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
    if (vec) {
        static PgcnPerDeviceOnce once;
        if (int rc = once.run([&]() -> int {
                PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
#ifdef PGCN_EXPERIMENTS
                PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
                PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
                PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_strip_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
#endif
                return PGCN_OK;
            }))
            return rc;
        const dim3 grid((unsigned)nwork, (unsigned)((f + 127) / 128));
#ifdef PGCN_EXPERIMENTS
        // measurement build only (tools/ab_build.sh exp -DPGCN_EXPERIMENTS): PGCN_STRIP_PROBE = 1 no compute phase,
        // 2 no panel staging, 4 phase timers instead of results
        static const int probe = getenv("PGCN_STRIP_PROBE") ? atoi(getenv("PGCN_STRIP_PROBE")) : 0;
        if (probe == 1) hipLaunchKernelGGL(spmm_strip_kernel<1>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws);
        else if (probe == 2) hipLaunchKernelGGL(spmm_strip_kernel<2>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws);
        else if (probe == 4) hipLaunchKernelGGL(spmm_strip_kernel<4>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws);
        else
#endif
        hipLaunchKernelGGL(spmm_strip_kernel<0>, grid, dim3(kThreads), kSmem, s, w4, r4, pairs, B, ldb, ncols, f, partial_ws);
    } else {
        hipLaunchKernelGGL(spmm_strip_generic_kernel, dim3((unsigned)nwork, (unsigned)((f + 31) / 32)), dim3(kThreads), 0, s,
                           w4, r4, pairs, B, ldb, ncols, f, partial_ws);
    }
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
