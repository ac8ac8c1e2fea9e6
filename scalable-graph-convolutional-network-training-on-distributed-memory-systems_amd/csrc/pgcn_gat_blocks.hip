// pgcn_gat_blocks.hip -- the dense 512 x 128 blocks of the attention pattern on the bf16 matrix cores (r06), fp32 accuracy.
//
// The gather kernels of the attention path (pgcn_spmm_heads.hip) pay one 1 KB row of the dense operand per stored entry and head
// set: 115 M entries = 118 GB through the vector L1 per pass, 7.3-7.9 ms, whatever the entry's neighbourhood looks like.  On a
// degree-sorted power-law graph most entries sit in a dense corner (the GCN path sends 43 % of them through 512 x 128 blocks at
// 3.8 ps per entry, pgcn_spmm_dense3.hip).  The attention weights are no stored values, but they are a FUNCTION of one number per
// row and one per column,
//     alpha_ij = exp(LeakyReLU(s1_i + s2_j) - m_i) / D_i        (rowstat[i] = (s1, m, 1 / D, 0): pgcn_gat_edge_softmax_f32, mode 0)
// so a block needs the PATTERN only (one bit per position: 8 KB per block instead of 256 KB of values) and its weights are computed in
// registers, in the A-operand order of v_mfma_f32_32x32x16_bf16, split into three bf16 planes (pgcn_bf16x3.h: exact) and multiplied
// with the split panel of the dense operand (the image of spmm_split_panels_kernel) -- six MFMAs per product as in the GCN blocks.
// Replaces, for the entries inside the blocks, the gather of /root/reference/GPU/PGAT.py:148 (`attention @ Z`) and of its autograd.
//
// FORWARD (rows i, columns j, B = Z):      out_i += sum_j alpha_ij Z_j,      V_i += sum_j c_ij Z_j,     C_i += sum_j c_ij
//     with c_ij = alpha_ij LeakyReLU'(s1_i + s2_j) -- the second accumulator of pgcn_spmm_heads_forward2_f32: two weight sets, two
//     chains of six MFMAs per accumulator block.
// BACKWARD (the transposed pattern: rows j, columns i, B = dOut):   dZ_j += sum_i alpha_ij dOut_i   and the edge gradient's row sums
//     ds2_j += sum_i c_ij (<dOut_i, Z_j> - t_i) = <U_j, Z_j> - sum_i c_ij t_i,     U_j = sum_i c_ij dOut_i
//     -- the SDDMM <dOut_i, Z_j> of pgcn_spmm_heads_grad_f32 becomes a second accumulator of the SAME product (linearity) and one dot
//     product per row at the end of a piece.  Row and column roles of the statistics swap: a row carries s2_j, a column
//     (s1_i, m_i, 1 / D_i, t_i).
//
// One workgroup of 8 waves = one piece (a run of blocks of one block row, pgcn_spmm_dense_bf16x3_f32's work list) x ONE head x one HALF of the
// 512 rows: wave w owns 32 rows x the head's 64 features x two weight sets = 4 accumulator blocks (64 registers), which leaves room for
// TWO sets of weight planes: the VALU work of a k step (8 weights: score, LeakyReLU, exp2, pattern bit, split -- ~190 issue slots) is as
// long as its 24 MFMAs, so the weights of step s + 1 are built in 24 slices UNDER the MFMAs of step s (first version, everything in
// sequence: 2.2 ms per pass for 6 029 blocks; see profiles/r06_gat_blocks.txt).  The second weight set is alpha restricted to the positive
// scores (an AND of the planes): c-weighted sums = slope x (all) + (1 - slope) x (positive), formed once per piece.
// Per quarter (32 columns) the head's half of the panel image (12 KB) comes into LDS by asynchronous copies, double buffered, one
// barrier per quarter; the per-column statistics of a block (128 x 1 or 4 floats) arrive the same way one block ahead.  Partial rows
// leave through slots (the caller runs pgcn_spmm_fixup_f32 over them, fixed order: deterministic).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "pgcn_bf16x3.h"
#include "pgcn_internal.h"
#include "pgcn_once.h"

#pragma clang diagnostic ignored "-Winline-asm"

// Measurement builds (PGCN_EXTRA_FLAGS=-DPGCN_GATB_PROBES, tools/probes_r06): the environment variable PGCN_GATB_PROBE then selects a
// TIMING-ONLY variant of the kernels (wrong sums): 1 = no MFMAs, 2 = no weight slices, 3 = neither (copies, barriers and LDS reads only).
// The library is built without the flag: one instantiation, probe 0.
namespace {
using namespace pgcn_bf16x3;

constexpr int kBR = PGCN_STRIP_TR;        // 512 rows per block
constexpr int kThreads = 512;             // 8 waves x 64 rows
static_assert(kBR == 8 * 64, "a wave owns 64 rows of a block");
constexpr int kD = 64;                    // features per head (two 32-column accumulator blocks)
constexpr int kQHead = 3 * 4 * kD * 16;   // a head's half of one quarter image: 3 planes x 4 k groups x 64 columns x 16 B = 12 KB
constexpr int kOffCol = 2 * kQHead;       // LDS: two quarter images, then two sets of column statistics
constexpr int kColBytes = 4 * kT * 4;     // one set: 4 components x 128 columns, fp32
constexpr size_t kSmem = kOffCol + 2 * kColBytes;
constexpr float kLog2e = 1.4426950408889634f;

struct BlockArgs {
    const int4 *work;             // {block row id, first block, number of blocks, first slot} per piece
    const int32_t *work_row0;     // first matrix row of a piece
    const int32_t *blk_img;       // a block's panel (index into panel_list = into the image)
    const int32_t *panel_list;    // first column of every panel
    const u32x4 *bits;            // pattern: [block][wave][lane] x 16 bytes, byte u = 2 ks + rb, bit 4 h + e (pgcn_hip.h)
    const char *image;            // split panels of B: [panel][feature block of 128] x 96 KB
    int32_t nfb;
    const float4 *rowstat;        // FWD: of the rows [nrows x KH];  BWD: of the columns [ncols x KH]
    const float *s2;              // FWD: of the columns [ncols x lds2];  BWD: of the rows [nrows x lds2]
    int64_t lds2;
    const float *t;               // BWD: [ncols x KH]
    const float *Z;               // BWD: the rows' own Z [nrows x ldz]
    int64_t ldz;
    int64_t nrows, ncols;
    int32_t KH;
    float slope;
    float *partial;               // slot rows of pw floats: FWD out;  BWD dZ | ds2 | 0
    int32_t pw;
    float *partial2;              // FWD: slot rows of pw2 floats: V | C | 0
    int32_t pw2;
};

// the head's half of quarter image `src` (this lane's 16 bytes of row w, and of row 8 + w for the first four waves) -> buffer `buf`
__device__ __forceinline__ void issue_quarter(const char *__restrict__ src, char *smem, int buf, int w, int lane) {
    __builtin_amdgcn_global_load_lds((gptr_t)(src + w * 2048 + lane * 16), (lptr_t)(smem + buf * kQHead + w * 1024), 16, 0, 0);
    if (w < 4)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (8 + w) * 2048 + lane * 16), (lptr_t)(smem + buf * kQHead + (8 + w) * 1024), 16, 0, 0);
}

// the three B reads (one per plane) of k step S, column block NB of the quarter
template <int S, int NB>
__device__ __forceinline__ void read_b(u32x4 (&bb)[3], uint32_t base) {
    lds_read_b128<0 * 4096 + S * 2048 + NB * 512>(bb[0], base);
    lds_read_b128<1 * 4096 + S * 2048 + NB * 512>(bb[1], base);
    lds_read_b128<2 * 4096 + S * 2048 + NB * 512>(bb[2], base);
}
// the column statistics of a lane's 8 columns of a k step (base: set, quarter, half wave, k step)
template <bool BWD>
__device__ __forceinline__ void read_cols(f32x4 (&cv)[8], uint32_t base) {
    lds_read_b128<0>(cv[0], base);
    lds_read_b128<16>(cv[1], base);
    if constexpr (BWD) {
        lds_read_b128<1 * 512>(cv[2], base);
        lds_read_b128<1 * 512 + 16>(cv[3], base);
        lds_read_b128<2 * 512>(cv[4], base);
        lds_read_b128<2 * 512 + 16>(cv[5], base);
        lds_read_b128<3 * 512>(cv[6], base);
        lds_read_b128<3 * 512 + 16>(cv[7], base);
    }
}
__device__ __forceinline__ void lds_wait_all(f32x4 (&cv)[8], u32x4 (&b0)[3], u32x4 (&b1)[3]) {
    // (one wait for everything outstanding: the compiler may place scalar loads -- the same counter -- anywhere between the reads)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]), "+v"(cv[4]), "+v"(cv[5]), "+v"(cv[6]), "+v"(cv[7]), "+v"(b0[0]),
                   "+v"(b0[1]), "+v"(b0[2]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b1[2]));
}

// What a lane carries from k step to k step: its row's statistics, the running sums, and the weights being built.
template <bool BWD>
struct Weights {
    float r_a, r_m, r_inv;       // FWD: s1, m, 1 / D of the row;  BWD: r_a = s2 of the row
    float slope;
    float sw, sp;                // sums of the weights (BWD: times t) over all / over the positive-score positions
    float x[8];                  // the 8 weights of the k step under construction
    uint32_t fm[8];              // ~0 where the raw score is positive
};

// Slice I (0..23) of the work that turns one k step's column statistics `cv` and pattern byte into the planes of alpha (An[0]) and
// of alpha restricted to the positive scores (An[1]): one MFMA's worth of VALU work each, so that the matrix pipe runs the PREVIOUS
// k step's 24 MFMAs underneath.  I = 0..15: element e = I / 2 (first half: score, LeakyReLU, exp2; second half: 1 / D, pattern bit,
// sums); I = 16..23: pair d = (I - 16) / 2 (first half: three-plane split; second half: the positive-score planes).
// (Selects through wave masks -- v_cmp, two wait states, v_cndmask -- are spelled as bit operations: no SGPR hazards in the slices.)
__device__ __forceinline__ uint32_t positive_mask(float raw) {          // ~0 where raw > 0 (the reference's LeakyReLU': x > 0 ? 1 : slope)
    int v;                                                               // as an integer: > 0 exactly for the positive floats (+0, -0: no)
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(v) : "v"(__builtin_bit_cast(int, raw)));
    return (uint32_t)(0 - v);
}
template <int E>
__device__ __forceinline__ uint32_t stored_mask(uint32_t byte) {        // ~0 where bit E of the pattern byte is set
    int v;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(v) : "v"(byte), "n"(E));
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t bit_select(uint32_t mask, uint32_t a, uint32_t b) {   // (mask & a) | (~mask & b)
    uint32_t v;                                                                            // (spelled out, the compiler expands it to three operations)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(v) : "v"(mask), "v"(a), "v"(b));
    return v;
}
__device__ __forceinline__ float and_bits(float x, uint32_t m) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & m); }
template <bool BWD, int I>
__device__ __forceinline__ void weight_slice(Weights<BWD> &W, const f32x4 (&cv)[8], uint32_t byte, u32x4 (&An)[2][3]) {
    if constexpr (I < 16) {
        constexpr int e = I >> 1, v = e >> 2, c = e & 3;
        if constexpr ((I & 1) == 0) {
            const float raw = W.r_a + cv[v][c];                                  // FWD: s1_i + s2_j;  BWD: s2_j + s1_i
            W.fm[e] = positive_mask(raw);
            const float r = __builtin_bit_cast(float, bit_select(W.fm[e], __builtin_bit_cast(uint32_t, raw), __builtin_bit_cast(uint32_t, raw * W.slope)));
            const float m = BWD ? cv[2 + v][c] : W.r_m;
            W.x[e] = __builtin_amdgcn_exp2f((r - m) * kLog2e);
            asm volatile("" : "+v"(W.x[e]), "+v"(W.fm[e]));                        // (the slice's work stays in the slice: see k_step)
        } else {
            const float inv = BWD ? cv[4 + v][c] : W.r_inv;
            const float xb = and_bits(W.x[e] * inv, stored_mask<e>(byte));        // zero where the position is not stored
            W.x[e] = xb;
            const float tt = BWD ? xb * cv[6 + v][c] : xb;
            W.sw += tt;
            W.sp += and_bits(tt, W.fm[e]);
            asm volatile("" : "+v"(W.x[e]), "+v"(W.sw), "+v"(W.sp));
        }
    } else {
        constexpr int d = (I - 16) >> 1;
        if constexpr ((I & 1) == 0) {
            uint32_t u1, u2, u3;
            split_pair(W.x[2 * d], W.x[2 * d + 1], u1, u2, u3);
            An[0][0][d] = u1; An[0][1][d] = u2; An[0][2][d] = u3;
            asm volatile("" : "+v"(An[0][0][d]), "+v"(An[0][1][d]), "+v"(An[0][2][d]));
        } else {
            const uint32_t pm = bit_select(0x0000ffffu, W.fm[2 * d], W.fm[2 * d + 1]);
            An[1][0][d] = An[0][0][d] & pm; An[1][1][d] = An[0][1][d] & pm; An[1][2][d] = An[0][2][d] & pm;
            asm volatile("" : "+v"(An[1][0][d]), "+v"(An[1][1][d]), "+v"(An[1][2][d]));
        }
    }
}

// One k step S of a quarter: its 24 MFMAs (accumulator [set][column block] += A[S][set] . B) with the weights of the NEXT k step built
// underneath, slice by slice.  After the first column block's MFMAs the B operands of step 1 are requested into the same registers.
template <bool BWD, int S, int PROBE>
__device__ __forceinline__ void k_step(f32x16 (&acc)[2][2], u32x4 (&A)[2][2][3], u32x4 (&bb)[2][3], Weights<BWD> &W, f32x4 (&cv)[8],
                                       uint32_t byte_next, uint32_t bbase) {
    PGCN_BF16X3_PRODUCTS;
    static_for<0, 24>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int nb = I / 12, i6 = (I % 12) / 2, set = I % 2;
        if constexpr (!(PROBE & 2)) weight_slice<BWD, I>(W, cv, byte_next, A[S ^ 1]);
        if constexpr (!(PROBE & 1)) acc[set][nb] = mfma_bf16(A[S][set][kPA[i6]], bb[nb][kPB[i6]], acc[set][nb]);
        if constexpr (S == 0 && I == 11) read_b<1, 0>(bb[0], bbase);
        if constexpr (S == 0 && I == 23) read_b<1, 1>(bb[1], bbase);
        __builtin_amdgcn_sched_barrier(0);                 // (one MFMA, one slice: left alone the scheduler clusters the MFMAs)
    });
}

template <bool BWD, int PROBE>
__global__ __launch_bounds__(kThreads, 2) void gat_blocks_kernel(const BlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4 wk = a.work[blockIdx.x];
    wk.y = __builtin_amdgcn_readfirstlane(wk.y); wk.z = __builtin_amdgcn_readfirstlane(wk.z);
    const int h = blockIdx.y >> 1, hf = blockIdx.y & 1, KH = a.KH;       // head, and which 256 rows of the 512-row blocks
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const int w8 = 4 * hf + (w >> 1), rb = w & 1;                        // the wave of 8 x 64 rows and the row block whose pattern bytes these are
    const int64_t row0 = (int64_t)a.work_row0[blockIdx.x] + 256 * hf + 32 * w;
    const int fb = h >> 1, half = h & 1;
    Weights<BWD> W;
    W.slope = a.slope; W.sw = W.sp = 0.f;
    {   // what a lane keeps of its row (row0 + lo; rows beyond the matrix carry no pattern bits)
        int64_t i = row0 + lo;
        i = i < a.nrows ? i : a.nrows - 1;
        if constexpr (BWD) {
            W.r_a = a.s2[i * a.lds2 + h];
            W.r_m = W.r_inv = 0.f;
        } else {
            const float4 q = a.rowstat[i * KH + h];
            W.r_a = q.x; W.r_m = q.y; W.r_inv = q.z;
        }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t rbase = lds0 + hi * 1024 + lo * 16;
    const uint32_t cbase = lds0 + kOffCol + hi * 32;
    const int nq = wk.z * 4;
    // a block's panel (image index) and first column, fetched ONE BLOCK AHEAD (a dependent load at the top of a quarter would stall it)
    // (loaded into vector registers and made scalar / used a quarter LATER, behind that quarter's own wait: no load is waited for where it is issued)
    auto panel_of = [&](int bi) -> int { return a.blk_img[(int64_t)wk.y + (bi < wk.z ? bi : wk.z - 1)]; };
    int pi_cur = __builtin_amdgcn_readfirstlane(panel_of(0)), pi_next = pi_cur;
    int pi_next_v = panel_of(1);
    int c0_v = a.panel_list[pi_cur];                     // (first use: the statistics of block 0)
    auto img_of = [&](int pi, int q) -> const char * {   // quarter q of the block whose panel is pi
        return a.image + ((int64_t)pi * a.nfb + fb) * (int64_t)kImgBytes + q * kQBytes + half * 1024;
    };
    // the per-column statistics of the block whose first column is c0 -> set `set`: wave w copies component w >> 1, columns 64 (w & 1) + lane
    auto issue_cols = [&](int set, int c0) {
        const int comp = w >> 1;
        if (!BWD && comp > 0) return;
        int64_t j = (int64_t)c0 + 64 * (w & 1) + lane;
        j = j < a.ncols ? j : a.ncols - 1;
        const float *src;
        if constexpr (BWD) src = comp < 3 ? reinterpret_cast<const float *>(a.rowstat + j * KH + h) + comp : a.t + j * KH + h;
        else src = a.s2 + j * a.lds2 + h;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + kOffCol + set * kColBytes + comp * 512 + (w & 1) * 256), 4, 0, 0);
    };
    auto bits_of = [&](int bi) -> u32x4 {
        bi = bi < wk.z ? bi : wk.z - 1;
        return a.bits[(((int64_t)wk.y + bi) * 8 + w8) * 64 + lane];
    };
    f32x16 acc[2][2];                                     // [alpha | alpha at the positive scores][column block]
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][nb][r] = 0.f;
    u32x4 A[2][2][3];                                     // planes of the current and of the next k step (parity of the step)
    u32x4 bb[2][3];
    f32x4 cv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- prologue: quarter 0 and the statistics of block 0 in LDS; the weights of k step 0 ------------------------------------
    issue_quarter(img_of(pi_cur, 0), smem, 0, w, lane);
    issue_cols(0, c0_v);
    u32x4 bits_cur = bits_of(0), bits_next = bits_of(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_cols<BWD>(cv, cbase);
    read_b<0, 0>(bb[0], rbase);
    read_b<0, 1>(bb[1], rbase);
    lds_wait_all(cv, bb[0], bb[1]);
    static_for<0, 24>([&](auto ic) { weight_slice<BWD, decltype(ic)::value>(W, cv, bits_cur.x >> (8 * rb), A[0]); });
    for (int g = 0; g < nq; ++g) {
        const int q = g & 3, bi = g >> 2;
        const uint32_t bbase = rbase + (g & 1) * kQHead;
        if (g > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copies of quarter g (and of the next block's statistics) have landed
            __syncthreads();                                   // ... everybody's have; nobody reads the other buffer (quarter g - 1) any more
            read_b<0, 0>(bb[0], bbase);
            read_b<0, 1>(bb[1], bbase);
        }
        // (requests whose operands were loaded a quarter ago come FIRST: the compiler's own wait for such an operand must not find this
        //  quarter's copies in flight)
        if (q == 1) {                                      // the next block's panel, requested at least a quarter ago; its first column is
            pi_next = __builtin_amdgcn_readfirstlane(pi_next_v);   // used a quarter from now
            c0_v = a.panel_list[pi_next];
        }
        if (q == 2) issue_cols((bi + 1) & 1, c0_v);        // (the set was last read in block bi - 1; it lands before the barrier of quarter 3)
        u32x4 bits_after = bits_next;                      // the block after the next one: requested here, taken over at the END of quarter 3
        int pi_after_v = pi_next_v;                        // (a load consumed where it is issued would be waited for with this quarter's copies)
        if (q == 3) {
            bits_after = bits_of(bi + 2);
            pi_after_v = panel_of(bi + 2);
        }
        // (the last quarter of the piece fetches the last block's quarter 0 once more instead of branching)
        issue_quarter(q == 3 ? img_of(pi_next, 0) : img_of(pi_cur, q + 1), smem, (g + 1) & 1, w, lane);
        // k step 0 of the quarter, the weights of k step 1 underneath
        const uint32_t wq = q == 0 ? bits_cur.x : q == 1 ? bits_cur.y : q == 2 ? bits_cur.z : bits_cur.w;   // units 4 q .. 4 q + 3
        const uint32_t cb = cbase + (bi & 1) * kColBytes + q * 128;
        read_cols<BWD>(cv, cb + 64);
        lds_wait_all(cv, bb[0], bb[1]);
        k_step<BWD, 0, PROBE>(acc, A, bb, W, cv, wq >> (16 + 8 * rb), bbase);
        // k step 1, the weights of the NEXT quarter's k step 0 underneath (the last quarter of the piece: no pattern, nothing is added)
        const uint32_t wn = q == 0 ? bits_cur.y : q == 1 ? bits_cur.z : q == 2 ? bits_cur.w : bits_next.x;
        const uint32_t cbn = cbase + ((q == 3 ? bi + 1 : bi) & 1) * kColBytes + ((q + 1) & 3) * 128;
        read_cols<BWD>(cv, cbn);
        lds_wait_all(cv, bb[0], bb[1]);
        k_step<BWD, 1, PROBE>(acc, A, bb, W, cv, g + 1 < nq ? wn >> (8 * rb) : 0u, bbase);
        if (q == 3) {
            bits_cur = bits_next;
            bits_next = bits_after;
            pi_cur = pi_next;
            pi_next_v = pi_after_v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the redundant last fetches must not outlive the workgroup's LDS)
    // ---- the piece's partial rows: alpha-weighted sums as they are, the c-weighted ones = slope (all) + (1 - slope) (positive) -----------
    const int F = KH * kD;
    const int64_t slot0 = (int64_t)wk.w + 256 * hf + 32 * w;
    const int colbase = h * kD + lo;
    const float sl = a.slope, om = 1.f - a.slope;
    float tot = sl * W.sw + om * W.sp;
    tot += __shfl_xor(tot, 32, 64);                       // both halves of the k steps: row lo
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
            a.partial[(slot0 + il) * a.pw + colbase + 32 * nb] = acc[0][nb][r];
            if constexpr (!BWD) a.partial2[(slot0 + il) * a.pw2 + colbase + 32 * nb] = sl * acc[0][nb][r] + om * acc[1][nb][r];
        }
    if constexpr (!BWD) {
        if (hi == 0) a.partial2[(slot0 + lo) * a.pw2 + F + h] = tot;
        if (h == 0 && hi == 0)                             // (the pad columns of a slot row are summed with the rest: zeros)
            for (int c = F + KH; c < a.pw2; ++c) a.partial2[(slot0 + lo) * a.pw2 + c] = 0.f;
    } else {
        // ds2 of row lo = <U, Z>_h - sum c t: the dot products in the accumulator layout (row il(r, hi), column lo), summed over the
        // 32 lanes of a half wave; then every lane picks the total of ITS row from the half that holds it
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
            int64_t i = row0 + il;
            i = i < a.nrows ? i : a.nrows - 1;
            const float *z = a.Z + i * a.ldz + colbase;
            p[r] = (sl * acc[0][0][r] + om * acc[1][0][r]) * z[0] + (sl * acc[0][1][r] + om * acc[1][1][r]) * z[32];
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] += __shfl_xor(p[r], o, 64);
        const int rsel = (lo & 3) + 4 * (lo >> 3), hsel = (lo >> 2) & 1;   // row lo = il(rsel, hsel)
        float mine = p[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mine = rsel == r ? p[r] : mine;
        const float other = __shfl_xor(mine, 32, 64);
        const float dot = hsel == hi ? mine : other;
        if (hi == 0) a.partial[(slot0 + lo) * a.pw + F + h] = dot - tot;
        if (h == 0 && hi == 0)
            for (int c = F + KH; c < a.pw; ++c) a.partial[(slot0 + lo) * a.pw + c] = 0.f;
    }
}

int launch_blocks(const char *who, bool bwd, const int32_t *work, int64_t nwork, const int32_t *work_row0, const int32_t *blk_img,
                  const uint32_t *bits, const int32_t *panel_list, int64_t npanels, const float *rowstat, const float *s2, int64_t lds2,
                  const float *t, const float *Z, int64_t ldz, float slope, int32_t heads, int32_t d, int64_t nrows, int64_t ncols,
                  const float *B, int64_t ldb, void *image_ws, int64_t image_ws_bytes, float *partial_ws, int64_t partial_ws_elems,
                  int64_t nslots, pgcn_stream_t stream) {
    const int64_t F = (int64_t)heads * d;
    if (nwork < 0 || npanels < 0 || heads <= 0 || nrows <= 0 || ncols <= 0 || ldb < F || lds2 < heads || nslots < 0)
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    if (d != kD || F > 256)
        return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "needs d = 64 and heads * d <= 256 (callers keep the gather kernels otherwise)");
    if (nwork == 0) return PGCN_OK;
    if (!work || !work_row0 || !blk_img || !bits || !panel_list || !rowstat || !s2 || !B || !image_ws || !partial_ws || npanels == 0 ||
        (bwd && (!t || !Z)))
        return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    if ((uintptr_t)work % 16 || (uintptr_t)bits % 16 || (uintptr_t)image_ws % 16 || (uintptr_t)rowstat % 16 || (bwd && ldz < F))
        return pgcn_set_error2(PGCN_EINVAL, who, "work / bits / image_ws / rowstat must be 16-byte aligned");
    const int64_t nfb = (F + kT - 1) / kT;
    if (image_ws_bytes < npanels * nfb * (int64_t)kImgBytes)
        return pgcn_set_error2(PGCN_ENOMEM, who, "panel image work-space too small");
    const int32_t pad = (int32_t)(F + (heads + 3) / 4 * 4);
    const int64_t per_slot = bwd ? pad : F + pad;
    if (partial_ws_elems < nslots * per_slot) return pgcn_set_error2(PGCN_ENOMEM, who, "partial work-space too small");
    if (nwork > 0x7fffffffLL || npanels > 0x7fffffffLL) return pgcn_set_error2(PGCN_EINVAL, who, "work / panel list too long");
    static PgcnPerDeviceOnce once;
    if (int rc = once.run([&]() -> int {
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)gat_blocks_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)gat_blocks_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
            return PGCN_OK;
        }))
        return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(spmm_split_panels_kernel, dim3((unsigned)npanels, (unsigned)nfb), dim3(kSplitThreads), 0, s, panel_list, B, ldb, ncols,
                       (int32_t)F, reinterpret_cast<u32x4 *>(image_ws));
    PGCN_HIP_CHECK(hipGetLastError());
    BlockArgs a{};
    a.work = reinterpret_cast<const int4 *>(work); a.work_row0 = work_row0; a.blk_img = blk_img; a.panel_list = panel_list;
    a.bits = reinterpret_cast<const u32x4 *>(bits); a.image = reinterpret_cast<const char *>(image_ws); a.nfb = (int32_t)nfb;
    a.rowstat = reinterpret_cast<const float4 *>(rowstat); a.s2 = s2; a.lds2 = lds2; a.t = t; a.Z = Z; a.ldz = ldz;
    a.nrows = nrows; a.ncols = ncols; a.KH = heads; a.slope = slope;
    a.partial = partial_ws; a.pw = bwd ? pad : (int32_t)F;
    a.partial2 = bwd ? nullptr : partial_ws + nslots * F; a.pw2 = pad;
    const dim3 grid((unsigned)nwork, 2 * (unsigned)heads), block(kThreads);
#ifdef PGCN_GATB_PROBES
    const char *pe = getenv("PGCN_GATB_PROBE");
    const int probe = pe ? atoi(pe) : 0;
#define PGCN_GATB_CASE(P)                                                                         \
    if (probe == P) {                                                                             \
        if (bwd) hipLaunchKernelGGL((gat_blocks_kernel<true, P>), grid, block, kSmem, s, a);      \
        else hipLaunchKernelGGL((gat_blocks_kernel<false, P>), grid, block, kSmem, s, a);         \
    }
    PGCN_GATB_CASE(1) PGCN_GATB_CASE(2) PGCN_GATB_CASE(3)
#undef PGCN_GATB_CASE
    if (probe >= 1 && probe <= 3) {
        PGCN_HIP_CHECK(hipGetLastError());
        return PGCN_OK;
    }
#endif
    if (bwd) hipLaunchKernelGGL((gat_blocks_kernel<true, 0>), grid, block, kSmem, s, a);
    else hipLaunchKernelGGL((gat_blocks_kernel<false, 0>), grid, block, kSmem, s, a);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

}  // namespace

extern "C" int pgcn_gat_blocks_forward_f32(const int32_t *work, int64_t nwork, const int32_t *work_row0, const int32_t *blk_img,
                                           const uint32_t *bits, const int32_t *panel_list, int64_t npanels, const float *rowstat,
                                           const float *s2, int64_t lds2, float slope, int32_t heads, int32_t d, int64_t nrows,
                                           int64_t ncols, const float *B, int64_t ldb, void *image_ws, int64_t image_ws_bytes,
                                           float *partial_ws, int64_t partial_ws_elems, int64_t nslots, pgcn_stream_t stream) {
    return launch_blocks("pgcn_gat_blocks_forward_f32", false, work, nwork, work_row0, blk_img, bits, panel_list, npanels, rowstat, s2, lds2,
                         nullptr, nullptr, 0, slope, heads, d, nrows, ncols, B, ldb, image_ws, image_ws_bytes, partial_ws, partial_ws_elems,
                         nslots, stream);
}

extern "C" int pgcn_gat_blocks_backward_f32(const int32_t *work, int64_t nwork, const int32_t *work_row0, const int32_t *blk_img,
                                            const uint32_t *bits, const int32_t *panel_list, int64_t npanels, const float *rowstat,
                                            const float *s2, int64_t lds2, const float *t, const float *Z, int64_t ldz, float slope,
                                            int32_t heads, int32_t d, int64_t nrows, int64_t ncols, const float *B, int64_t ldb,
                                            void *image_ws, int64_t image_ws_bytes, float *partial_ws, int64_t partial_ws_elems,
                                            int64_t nslots, pgcn_stream_t stream) {
    return launch_blocks("pgcn_gat_blocks_backward_f32", true, work, nwork, work_row0, blk_img, bits, panel_list, npanels, rowstat, s2, lds2, t,
                         Z, ldz, slope, heads, d, nrows, ncols, B, ldb, image_ws, image_ws_bytes, partial_ws, partial_ws_elems, nslots, stream);
}
