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
// One workgroup of 8 waves = one piece (a run of blocks of one block row, pgcn_spmm_dense_bf16x3_f32's work list) x ONE head: wave w owns
// rows [64 w, 64 w + 64) as two 32-row blocks x the head's 64 features x two weight sets = 8 accumulator blocks (128 registers).
// Per quarter (32 columns) the head's half of the panel image (12 KB) comes into LDS by asynchronous copies, double buffered, one
// barrier per quarter; the per-column statistics of a block (128 x 1 or 4 floats) arrive the same way one block ahead.  Partial rows
// leave through slots (the caller runs pgcn_spmm_fixup_f32 over them, fixed order: deterministic).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pgcn_bf16x3.h"
#include "pgcn_internal.h"
#include "pgcn_once.h"

#pragma clang diagnostic ignored "-Winline-asm"

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

template <bool BWD>
__global__ __launch_bounds__(kThreads, 2) void gat_blocks_kernel(const BlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int4 wk = a.work[blockIdx.x];
    wk.y = __builtin_amdgcn_readfirstlane(wk.y); wk.z = __builtin_amdgcn_readfirstlane(wk.z);
    const int h = blockIdx.y, KH = a.KH;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, lo = lane & 31;
    const int64_t row0 = (int64_t)a.work_row0[blockIdx.x] + 64 * w;
    const int fb = h >> 1, half = h & 1;
    const float slope = a.slope;
    // ---- what a lane keeps of its two rows (row = row0 + 32 rb + lo; rows beyond the matrix carry no pattern bits) ----------------
    float r_a[2], r_m[2], r_inv[2];              // FWD: s1, m, 1 / D;  BWD: r_a = s2 of the row
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        int64_t i = row0 + 32 * rb + lo;
        i = i < a.nrows ? i : a.nrows - 1;
        if constexpr (BWD) {
            r_a[rb] = a.s2[i * a.lds2 + h];
            r_m[rb] = r_inv[rb] = 0.f;
        } else {
            const float4 q = a.rowstat[i * KH + h];
            r_a[rb] = q.x; r_m[rb] = q.y; r_inv[rb] = q.z;
        }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const uint32_t rbase = lds0 + hi * 1024 + lo * 16;
    const uint32_t cbase = lds0 + kOffCol + hi * 32;
    const int nq = wk.z * 4;
    auto img_of = [&](int g) -> const char * {           // quarter g of the piece (clamped: the last quarters fetch themselves again)
        g = g < nq ? g : nq - 1;
        const int64_t pi = a.blk_img[(int64_t)wk.y + (g >> 2)];
        return a.image + (pi * a.nfb + fb) * (int64_t)kImgBytes + (g & 3) * kQBytes + half * 1024;
    };
    // the per-column statistics of block bi -> set bi & 1: wave w copies component w >> 1, columns 64 (w & 1) + lane
    auto issue_cols = [&](int bi) {
        bi = bi < wk.z ? bi : wk.z - 1;
        const int comp = w >> 1;
        if (!BWD && comp > 0) return;
        int64_t j = (int64_t)a.panel_list[a.blk_img[(int64_t)wk.y + bi]] + 64 * (w & 1) + lane;
        j = j < a.ncols ? j : a.ncols - 1;
        const float *src;
        if constexpr (BWD) src = comp < 3 ? reinterpret_cast<const float *>(a.rowstat + j * KH + h) + comp : a.t + j * KH + h;
        else src = a.s2 + j * a.lds2 + h;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + kOffCol + (bi & 1) * kColBytes + comp * 512 + (w & 1) * 256), 4, 0, 0);
    };
    auto bits_of = [&](int bi) -> u32x4 {
        bi = bi < wk.z ? bi : wk.z - 1;
        return a.bits[(((int64_t)wk.y + bi) * 8 + w) * 64 + lane];
    };
    f32x16 acc[2][2][2];                                  // [weight set: alpha, c][row block][column block]
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][rb][nb][r] = 0.f;
    float csum[2] = {0.f, 0.f};                           // FWD: sum of c;  BWD: sum of c t   (this lane's 8 of the 16 columns of a k step)
    issue_quarter(img_of(0), smem, 0, w, lane);
    issue_cols(0);
    u32x4 bits_cur = bits_of(0), bits_next = bits_cur;
    PGCN_BF16X3_PRODUCTS;
    for (int g = 0; g < nq; ++g) {
        const int q = g & 3, bi = g >> 2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's copies of quarter g (and of the block's column statistics) have landed
        __syncthreads();                                   // ... everybody's have; nobody reads the other buffer (quarter g - 1) any more
        issue_quarter(img_of(g + 1), smem, (g + 1) & 1, w, lane);
        if (q == 2) issue_cols(bi + 1);                    // (its set was last read in block bi - 1; it lands before the barrier of quarter 3)
        if (q == 0) {
            bits_cur = bits_next;
            bits_next = bits_of(bi + 1);
        }
        const uint32_t bw = q == 0 ? bits_cur.x : q == 1 ? bits_cur.y : q == 2 ? bits_cur.z : bits_cur.w;   // units 4 q .. 4 q + 3
        const uint32_t bbase = rbase + (g & 1) * kQHead;
        const uint32_t cb = cbase + (bi & 1) * kColBytes + q * 128;
        static_for<0, 2>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // the column statistics of this lane's 8 columns k = 32 q + 16 s + 8 hi + e
            f32x4 cv[8];
            lds_read_b128<0 * 512 + s * 64>(cv[0], cb);
            lds_read_b128<0 * 512 + s * 64 + 16>(cv[1], cb);
            if constexpr (BWD) {
                lds_read_b128<1 * 512 + s * 64>(cv[2], cb);
                lds_read_b128<1 * 512 + s * 64 + 16>(cv[3], cb);
                lds_read_b128<2 * 512 + s * 64>(cv[4], cb);
                lds_read_b128<2 * 512 + s * 64 + 16>(cv[5], cb);
                lds_read_b128<3 * 512 + s * 64>(cv[6], cb);
                lds_read_b128<3 * 512 + s * 64 + 16>(cv[7], cb);
            } else {
                cv[2] = cv[3] = cv[4] = cv[5] = cv[6] = cv[7] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            u32x4 bb[2][3];
            read_b<s, 0>(bb[0], bbase);
            read_b<s, 1>(bb[1], bbase);
            // (one wait for the step's reads: the compiler may place scalar loads -- the same counter -- anywhere between them)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]), "+v"(cv[4]), "+v"(cv[5]), "+v"(cv[6]), "+v"(cv[7]), "+v"(bb[0][0]),
                           "+v"(bb[0][1]), "+v"(bb[0][2]), "+v"(bb[1][0]), "+v"(bb[1][1]), "+v"(bb[1][2]));
            u32x4 aw[2][3], ac[2][3];                       // the planes of alpha and of c, both row blocks
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const uint32_t byte = bw >> (8 * (2 * s + rb));
                float wv[8], cw[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float colA = cv[e >> 2][e & 3];                     // FWD: s2_j;  BWD: s1_i
                    const float raw = r_a[rb] + colA;
                    const bool pos = raw > 0.f;
                    const float r = pos ? raw : raw * slope;
                    const float m = BWD ? cv[2 + (e >> 2)][e & 3] : r_m[rb];
                    const float inv = BWD ? cv[4 + (e >> 2)][e & 3] : r_inv[rb];
                    float x = __builtin_amdgcn_exp2f((r - m) * kLog2e) * inv;
                    x = (byte >> e) & 1u ? x : 0.f;
                    wv[e] = x;
                    cw[e] = pos ? x : x * slope;
                    if constexpr (BWD) csum[rb] = fmaf(cw[e], cv[6 + (e >> 2)][e & 3], csum[rb]);
                    else csum[rb] += cw[e];
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t u1, u2, u3;
                    split_pair(wv[2 * d], wv[2 * d + 1], u1, u2, u3);
                    aw[rb][0][d] = u1; aw[rb][1][d] = u2; aw[rb][2][d] = u3;
                    split_pair(cw[2 * d], cw[2 * d + 1], u1, u2, u3);
                    ac[rb][0][d] = u1; ac[rb][1][d] = u2; ac[rb][2][d] = u3;
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    acc[0][rb][0] = mfma_bf16(aw[rb][kPA[i]], bb[0][kPB[i]], acc[0][rb][0]);
                    acc[1][rb][0] = mfma_bf16(ac[rb][kPA[i]], bb[0][kPB[i]], acc[1][rb][0]);
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    acc[0][rb][1] = mfma_bf16(aw[rb][kPA[i]], bb[1][kPB[i]], acc[0][rb][1]);
                    acc[1][rb][1] = mfma_bf16(ac[rb][kPA[i]], bb[1][kPB[i]], acc[1][rb][1]);
                }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the redundant last fetches must not outlive the workgroup's LDS)
    // ---- the piece's partial rows ---------------------------------------------------------------------------------------------
    const int F = KH * kD;
    const int64_t slot0 = (int64_t)wk.w + 64 * w;
    const int colbase = h * kD + lo;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                a.partial[(slot0 + 32 * rb + il) * a.pw + colbase + 32 * nb] = acc[0][rb][nb][r];
                if constexpr (!BWD) a.partial2[(slot0 + 32 * rb + il) * a.pw2 + colbase + 32 * nb] = acc[1][rb][nb][r];
            }
    float tot[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) tot[rb] = csum[rb] + __shfl_xor(csum[rb], 32, 64);     // both halves of the k steps: row 32 rb + lo
    if constexpr (!BWD) {
        if (hi == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) a.partial2[(slot0 + 32 * rb + lo) * a.pw2 + F + h] = tot[rb];
        }
        if (h == 0 && hi == 0) {                           // (the pad columns of a slot row are summed with the rest: zeros)
            for (int c = F + KH; c < a.pw2; ++c)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) a.partial2[(slot0 + 32 * rb + lo) * a.pw2 + c] = 0.f;
        }
    } else {
        // ds2 of row 32 rb + lo = <U, Z>_h - sum c t: the dot products in the accumulator layout (row il(r, hi), column lo), summed over
        // the 32 lanes of a half wave; then every lane picks the total of ITS row from the half that holds it
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = (r & 3) + 8 * (r >> 2) + 4 * hi;
                int64_t i = row0 + 32 * rb + il;
                i = i < a.nrows ? i : a.nrows - 1;
                const float *z = a.Z + i * a.ldz + colbase;
                p[r] = acc[1][rb][0][r] * z[0] + acc[1][rb][1][r] * z[32];
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
            if (hi == 0) a.partial[(slot0 + 32 * rb + lo) * a.pw + F + h] = dot - tot[rb];
        }
        if (h == 0 && hi == 0) {
            for (int c = F + KH; c < a.pw; ++c)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) a.partial[(slot0 + 32 * rb + lo) * a.pw + c] = 0.f;
        }
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
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)gat_blocks_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
            PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)gat_blocks_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmem));
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
    const dim3 grid((unsigned)nwork, (unsigned)heads), block(kThreads);
    if (bwd) hipLaunchKernelGGL(gat_blocks_kernel<true>, grid, block, kSmem, s, a);
    else hipLaunchKernelGGL(gat_blocks_kernel<false>, grid, block, kSmem, s, a);
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
