// pgcn_dense.hip -- the dense products of a layer on the bf16 matrix cores at fp32 accuracy, fused with what surrounds them:
//     forward    Y  = relu(X . W^T)                  (/root/reference/GPU/PGCN.py:146-147  `F.relu(self.linear(AH))`)
//     backward   Gm = G (.) [Y > 0],  dX = Gm . W    (autograd of the same two lines; dW = Gm^T . X stays a library GEMM)
// north_star reserves MFMA for exactly this contraction; rounds 1-4 ran it as stock rocBLAS kernels (85 us per product at
// n = 232 965, f = 128: profiles/r04_gemm_pick.txt) plus a ReLU pass (35 us) and a mask pass (50 us).  An n x K x N product with
// K, N <= 128 moves 2 x n x 512 B and needs 2 n K N flops: HBM-bound (30 us at 8 TB/s for the benchmark layer) as long as the
// matrix pipes stay under that, which the fp32 MFMA does not (49 us at its peak) and six bf16 MFMAs do (18 us).
//
// Arithmetic: the three-plane bf16 split of csrc/pgcn_spmm_dense3.hip (x = x1 + x2 + x3 exactly, six partial products smallest
// first, fp32 accumulation inside the MFMA): error class of an fp32 dot product, deterministic, no dependence on the grid.
// Not bit-identical to a library GEMM (neither are two library kernels to each other); tests hold it to 1e-6 of sum |x||w|.
//
// Layout.  ONE persistent workgroup per CU, 8 waves.  W is split ONCE per workgroup into the B-operand image in LDS
//     image[plane p][k step ks][column block nb][lane] x 16 B   (8 bf16: Bm[16 ks + 8 (lane >> 5) + j][32 nb + (lane & 31)])
// = 96 KB for 128 x 128, a lane's operand of v_mfma_f32_32x32x16_bf16 is one slot, a wave reads 1 KB contiguous (no bank
// conflicts).  Bm = W^T (forward: Bm[k][o] = W[o][k]) or W (backward: Bm[o][k] = W[o][k]).  A wave owns 32-row tiles of X
// (tile t = 8 workgroup + wave, stride 8 workgroups): its lane (lo, hi) reads the A operands straight from global memory --
// row lo, columns 16 ks + 8 hi .. + 8 as two 16-byte loads per k step (the whole 32 x 128 tile is 16 loads in flight per
// lane; every byte of X is read once, a 128-byte line is touched by four loads issued back to back) --, splits them in
// registers and runs nb x 6 MFMAs per k step against the image.  The forward keeps the next tile's loads in flight under the
// second half of the MFMAs of the current one (the other wave of the SIMD covers the rest); the backward loads G and Y, keeps G (.) [Y > 0] and
// writes it out as Gm for the weight gradient.  C leaves as 16 x nb dword stores per lane, two full 128-byte lines per store.
//
// The index arithmetic lives in functions that a host build of this same file runs lane by lane with an emulated MFMA
// (tests/test_dense_fused.py, -DPGCN_DENSE_HOST_EMU with clang++): image slots, operand lanes and the accumulator layout
// are checked on the CPU against numpy; the MFMA operand layout itself is the one pgcn_spmm_dense3.hip runs on hardware.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#ifdef PGCN_DENSE_HOST_EMU
#define PG_HD inline
#else
#include <hip/hip_runtime.h>
#define PG_HD __device__ __forceinline__
#endif

namespace pgcn_dense {

// Measurement builds (tools/micro/build_dense_fused_bench.sh compiles this file again with -DPGCN_DENSE_PREFETCH=0 / 2 and other
// entry-point names): where the forward kernel issues the next tile's loads.  The library is built with 1.
#ifndef PGCN_DENSE_PREFETCH
#define PGCN_DENSE_PREFETCH 1
#endif
constexpr int kPrefetch = PGCN_DENSE_PREFETCH;
// ... and whether the masked (backward) kernel streams its operand by half tiles (1) or loads a whole tile, then multiplies (0)
#ifndef PGCN_DENSE_MASK_PIPE
#define PGCN_DENSE_MASK_PIPE 1
#endif
constexpr bool kMaskPipe = PGCN_DENSE_MASK_PIPE != 0;
// ... and (1) the B operands of step t + 1 read from LDS under the MFMAs of step t, tiles in two register sets that swap roles
// (no copy, so no wait for the tile's stores), or (0) the first version: LDS reads, wait, MFMAs, per step; cur = nxt per tile
#ifndef PGCN_DENSE_PIPE
#define PGCN_DENSE_PIPE 1
#endif
constexpr bool kPipe = PGCN_DENSE_PIPE != 0;
// ... candidates prepared at the end of r04, not yet run on hardware (the library keeps them off until they have been):
//   PGCN_DENSE_FASTPATH 1: tiles that lie inside the matrix (all but a wave's last) load and store without per-piece predicates
//     (the first version wraps each of its 16 loads and 64 stores in an exec-mask branch);
//   PGCN_DENSE_SPREAD 1: a wave's tiles are numbered so that the last, partial round is spread over all CUs;
//   PGCN_DENSE_NT_STORE 1: the stores of C carry the non-temporal hint (C is not read again by this kernel);
//   PGCN_DENSE_CT 1: the MFMA computes the TRANSPOSED tile (the image of W as the A operand, the rows of X as B -- both operands
//     have the same lane layout, so only the two arguments swap): a lane then holds 4 x 4 CONSECUTIVE columns of ONE row of C
//     and stores 16 bytes at a time, 16 store instructions per tile instead of 64, addressed like its loads;
// and TIMING-ONLY probes (wrong results by construction; tools/micro/dense_fused_bench labels them):
//   PGCN_DENSE_PROBE 1: no MFMAs;  2: no stores of C (one never-true predicate over all accumulators keeps the products alive);
//   3: the wave's first tile is multiplied again and again (no loads after the first; registers made opaque per tile).
#ifndef PGCN_DENSE_FASTPATH
#define PGCN_DENSE_FASTPATH 0
#endif
#ifndef PGCN_DENSE_NT_STORE
#define PGCN_DENSE_NT_STORE 0
#endif
#ifndef PGCN_DENSE_PROBE
#define PGCN_DENSE_PROBE 0
#endif
#ifndef PGCN_DENSE_CT
#define PGCN_DENSE_CT 0
#endif
#ifndef PGCN_DENSE_SPREAD
#define PGCN_DENSE_SPREAD 0
#endif
constexpr bool kSpread = PGCN_DENSE_SPREAD != 0;
constexpr bool kFastPath = PGCN_DENSE_FASTPATH != 0, kNtStore = PGCN_DENSE_NT_STORE != 0, kCT = PGCN_DENSE_CT != 0;
constexpr int kProbe = PGCN_DENSE_PROBE;
constexpr int kRows = 32;                 // rows of a wave's tile = M of the MFMA
constexpr int kMaxF = 128;                // K and N of a product
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kSlotsPerPlane = 8 * 4 * 64;                 // (k step, column block, lane)
constexpr int kPlaneBytes = kSlotsPerPlane * 16;           // 32 KB
constexpr int kImageBytes = 3 * kPlaneBytes;               // 96 KB

#include "pgcn_dense_common.h"

// byte offset of a lane's B operand in the image
PG_HD int image_offset(int plane, int ks, int nb, int lane) { return plane * kPlaneBytes + ((ks * 4 + nb) * 64 + lane) * 16; }

// Slot s (0 .. kSlotsPerPlane) of the image, all three planes: the eight values Bm[16 ks + 8 hi + j][32 nb + lo].
// transposed = 1: Bm = W^T (W is N x K, row-major, ldw); 0: Bm = W (W is K x N).  Outside K x N: zeros.
// In two steps, so that a thread's loads of all its slots are in flight together (branch-free: clamped addresses, the value
// dropped afterwards -- the first version loaded element by element under a branch: 32 dependent L2 round trips per
// workgroup before the first tile, ~20 us of an 88 us launch).
PG_HD void slot_load(const float *__restrict__ W, int64_t ldw, int transposed, int K, int N, int s, float (&v)[8]) {
    const int lane = s & 63, nb = (s >> 6) & 3, ks = s >> 8;
    const int col = 32 * nb + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    const int colc = col < N ? col : N - 1;
    const int64_t sk = transposed ? 1 : ldw, sc = transposed ? ldw : 1;      // (one address, one load: no branch on the mode)
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j, kc = k < K ? k : K - 1;
        x[j] = W[kc * sk + colc * sc];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (k0 + j < K && col < N) ? x[j] : 0.f;
}
PG_HD void slot_store(char *image, int s, const float (&v)[8]) {
    const int lane = s & 63, nb = (s >> 6) & 3, ks = s >> 8;
    u32x4 p[3];
    const f32x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
    split8(lo4, hi4, p);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(image + image_offset(pl, ks, nb, lane)) = p[pl];
}

// A lane's 16-byte pieces of a tile: piece (ks, h) = A[row0 + lo][16 ks + 8 hi + 4 h .. + 4]; zeros outside n x K (K % 4 == 0).
struct Piece {
    int64_t off;      // element offset from the matrix base (valid only when ok)
    bool ok;
};
PG_HD Piece piece_of(int64_t row0, int64_t n, int K, int64_t ld, int lane, int ks, int h) {
    const int64_t row = row0 + (lane & 31);
    const int k = 16 * ks + 8 * (lane >> 5) + 4 * h;
    Piece p;
    p.ok = row < n && k < K;
    p.off = row * ld + k;
    return p;
}
// threshold_backward(g, y, 0): the gradient where y > 0 (NaN in y keeps it, like ATen's `y <= 0 ? 0 : g`)
PG_HD f32x4 mask4(const f32x4 &g, const f32x4 &y) {
    f32x4 r;
    r.x = y.x <= 0.f ? 0.f : g.x; r.y = y.y <= 0.f ? 0.f : g.y; r.z = y.z <= 0.f ? 0.f : g.z; r.w = y.w <= 0.f ? 0.f : g.w;
    return r;
}
PG_HD float relu1(float x) { return x < 0.f ? 0.f : x; }                      // clamp_min(0): NaN stays NaN

// accumulator register r of lane (lo, hi), column block nb -> element (row0 + (r & 3) + 8 (r >> 2) + 4 hi, 32 nb + lo) of C
PG_HD void store1(float *p, float x) {
#ifndef PGCN_DENSE_HOST_EMU
    if constexpr (kNtStore) {
        __builtin_nontemporal_store(x, p);
        return;
    }
#endif
    *p = x;
}
// ... and of the transposed tile (kCT): register r = 4 q + e of lane (lo, hi), block nb -> element (row0 + lo, 32 nb + 8 q + 4 hi + e)
PG_HD void store_ct(const f32x16 *acc, int nblk, float *C, int64_t ldc, int64_t row0, int64_t n, int N, int lane, int relu) {
    const int hi = lane >> 5, lo = lane & 31;
    const int64_t row = row0 + lo;
    if (row >= n) return;
    float *dst = C + row * ldc;
    const bool wide = ldc % 4 == 0 && (uintptr_t)C % 16 == 0 && N % 4 == 0;      // (uniform) whole, aligned 16-byte stores
    if (wide) {
#pragma unroll
        for (int nb = 0; nb < nblk; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * nb + 8 * q + 4 * hi;
                f32x4 v = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
                if (relu) { v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w); }
                if (col < N) *reinterpret_cast<f32x4 *>(dst + col) = v;
            }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < nblk; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int col = 32 * nb + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (col < N) dst[col] = relu ? relu1(acc[nb][r]) : acc[nb][r];
        }
}
PG_HD void store_c(const f32x16 *acc, int nblk, float *C, int64_t ldc, int64_t row0, int64_t n, int N, int lane, int relu) {
    const int hi = lane >> 5, lo = lane & 31;
    if constexpr (kCT) {
        store_ct(acc, nblk, C, ldc, row0, n, N, lane, relu);
        return;
    }
#ifndef PGCN_DENSE_HOST_EMU
    if constexpr (kProbe == 2) {                      // timing only: every accumulator is needed, nothing is written
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < nblk; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[nb][r];
        if (sum == 123456.789f) C[row0 * ldc + lo] = sum;
        return;
    }
#endif
    if constexpr (kFastPath) {
        if (row0 + kRows <= n && 32 * nblk == N) {    // (wave-uniform) the tile lies inside C: no predicates
            float *base = C + (row0 + 4 * hi) * ldc + lo;
#pragma unroll
            for (int nb = 0; nb < nblk; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    store1(base + (int64_t)((r & 3) + 8 * (r >> 2)) * ldc + 32 * nb, relu ? relu1(acc[nb][r]) : acc[nb][r]);
            return;
        }
    }
#pragma unroll
    for (int nb = 0; nb < nblk; ++nb) {
        const int col = 32 * nb + lo;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < n) {
#ifndef PGCN_DENSE_HOST_EMU
                if constexpr (kNtStore) {
                    store1(&C[row * ldc + col], relu ? relu1(acc[nb][r]) : acc[nb][r]);
                    continue;
                }
#endif
                C[row * ldc + col] = relu ? relu1(acc[nb][r]) : acc[nb][r];
            }
        }
    }
}

// ---- argument checks and the error string (both builds) ---------------------------------------------------------------------
thread_local char g_err[256] = "";
int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

int check(const void *A, int64_t lda, int64_t n, int K, int N, const void *W, int64_t ldw, int wrows, int wcols, const void *C,
          int64_t ldc) {
    if (n < 0 || K <= 0 || N <= 0 || !W || (n > 0 && (!A || !C))) return fail(-1, "pgcn_dense: bad argument");
    if (K > kMaxF || N > kMaxF) return fail(-2, "pgcn_dense: widths above 128 are left to the library GEMM");
    if (K % 4 || lda % 4 || (uintptr_t)A % 16) return fail(-2, "pgcn_dense: rows of the left operand must be 16-byte pieces");
    if (lda < K || ldc < N || ldw < wcols || wrows <= 0) return fail(-1, "pgcn_dense: leading dimension below the width");
    if (n > ((int64_t)1 << 40)) return fail(-1, "pgcn_dense: n out of range");
    return 0;
}


// ---- a wave's operand tiles (both builds: the host build runs these lane by lane) -----------------------------------------------
template <int NKS>
struct TileA {
    f32x4 v[NKS][2];
};

template <int NKS>
PG_HD void load_tile(TileA<NKS> &t, const float *__restrict__ A, int64_t lda, int64_t row0, int64_t n, int K, int lane) {
    if constexpr (kFastPath) {
        if (row0 + kRows <= n && 16 * NKS == K) {     // (wave-uniform) the tile lies inside A: 16 loads off one address
            const float *base = A + (row0 + (lane & 31)) * lda + 8 * (lane >> 5);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h) t.v[ks][h] = *reinterpret_cast<const f32x4 *>(base + 16 * ks + 4 * h);
            return;
        }
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const Piece p = piece_of(row0, n, K, lda, lane, ks, h);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            t.v[ks][h] = p.ok ? *reinterpret_cast<const f32x4 *>(A + p.off) : z;
        }
}
// G (.) [Y > 0], written out as Gm (when asked for) on the way
template <int NKS>
PG_HD void load_tile_masked(TileA<NKS> &t, const float *__restrict__ G, int64_t ldg, const float *__restrict__ Y, int64_t ldy,
                            float *__restrict__ Gm, int64_t ldgm, int64_t row0, int64_t n, int K, int lane) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const Piece pg = piece_of(row0, n, K, ldg, lane, ks, h);
            const Piece py = piece_of(row0, n, K, ldy, lane, ks, h);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            t.v[ks][h] = pg.ok ? *reinterpret_cast<const f32x4 *>(G + pg.off) : z;
            const f32x4 y = py.ok ? *reinterpret_cast<const f32x4 *>(Y + py.off) : z;
            t.v[ks][h] = mask4(t.v[ks][h], y);
        }
    if (Gm) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const Piece pm = piece_of(row0, n, K, ldgm, lane, ks, h);
                if (pm.ok) *reinterpret_cast<f32x4 *>(Gm + pm.off) = t.v[ks][h];
            }
    }
}

// ---- the masked operand as a stream of HALF tiles (k steps [KS0, KS0 + CNT) of a tile) ---------------------------------------
// G and Y of a half are 2 x CNT x 2 loads of 16 bytes; while one half is multiplied the next one (the second half of the tile,
// or the first half of the wave's next tile) is in flight: 32 + 64 + 64 registers instead of the 128 + 64 a whole tile of
// G and Y would hold beside the accumulators.
template <int CNT>
struct HalfRaw {
    f32x4 g[CNT][2], y[CNT][2];
};
template <int CNT>
PG_HD void load_half(HalfRaw<CNT> &r, const float *__restrict__ G, int64_t ldg, const float *__restrict__ Y, int64_t ldy,
                     int64_t row0, int64_t n, int K, int lane, int ks0) {
    if constexpr (kFastPath) {
        if (row0 + kRows <= n && 32 * CNT == K) {     // (wave-uniform) inside the matrix: no predicates
            const int64_t at = 8 * (lane >> 5) + 16 * ks0;
            const float *gb = G + (row0 + (lane & 31)) * ldg + at, *yb = Y + (row0 + (lane & 31)) * ldy + at;
#pragma unroll
            for (int i = 0; i < CNT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    r.g[i][h] = *reinterpret_cast<const f32x4 *>(gb + 16 * i + 4 * h);
                    r.y[i][h] = *reinterpret_cast<const f32x4 *>(yb + 16 * i + 4 * h);
                }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const Piece pg = piece_of(row0, n, K, ldg, lane, ks0 + i, h);
            const Piece py = piece_of(row0, n, K, ldy, lane, ks0 + i, h);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            r.g[i][h] = pg.ok ? *reinterpret_cast<const f32x4 *>(G + pg.off) : z;
            r.y[i][h] = py.ok ? *reinterpret_cast<const f32x4 *>(Y + py.off) : z;
        }
}
template <int CNT>
PG_HD void mask_half(f32x4 (&v)[CNT][2], const HalfRaw<CNT> &r, float *__restrict__ Gm, int64_t ldgm, int64_t row0, int64_t n, int K,
                     int lane, int ks0) {
    if constexpr (kFastPath) {
        if (row0 + kRows <= n && 32 * CNT == K) {
            float *mb = Gm ? Gm + (row0 + (lane & 31)) * ldgm + 8 * (lane >> 5) + 16 * ks0 : nullptr;
#pragma unroll
            for (int i = 0; i < CNT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v[i][h] = mask4(r.g[i][h], r.y[i][h]);
                    if (mb) *reinterpret_cast<f32x4 *>(mb + 16 * i + 4 * h) = v[i][h];
                }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v[i][h] = mask4(r.g[i][h], r.y[i][h]);
            if (Gm) {
                const Piece pm = piece_of(row0, n, K, ldgm, lane, ks0 + i, h);
                if (pm.ok) *reinterpret_cast<f32x4 *>(Gm + pm.off) = v[i][h];
            }
        }
}
#ifndef PGCN_DENSE_HOST_EMU
// ---- device ---------------------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// one partial product: x = a plane of the streamed operand (rows of X / G), w = a plane of the image
PG_HD f32x16 mma(const u32x4 &x, const u32x4 &w, const f32x16 &c) {
    if constexpr (kCT) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), c, 0, 0, 0);
}
PG_HD void read_b(u32x4 (&b)[3], const char *image, int ks, int nb, int lane) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(image + image_offset(pl, ks, nb, lane));
}
// CNT k steps (ks0 ..) x NBLK column blocks as one sequence of steps t = i NBLK + nb; the operands of step t + 1 are requested
// before the six MFMAs of step t (LDS returns in order: the wait before a step leaves the newest three reads outstanding)
template <int NBLK, int CNT>
PG_HD void product_steps(const f32x4 (&v)[CNT][2], const char *image, int lane, int ks0, f32x16 (&acc)[NBLK]) {
    constexpr int T = CNT * NBLK;
    u32x4 b[2][3], a[3];
    read_b(b[0], image, ks0, 0, lane);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        PGCN_DENSE_PRODUCTS;
        const int i = t / NBLK, nb = t % NBLK;
        if (t + 1 < T) read_b(b[(t + 1) & 1], image, ks0 + (t + 1) / NBLK, (t + 1) % NBLK, lane);
        if (nb == 0) split8(v[i][0], v[i][1], a);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if constexpr (kProbe == 1) {              // timing only: the operands stay needed, the matrix pipe stays idle
                acc[nb][j] += __builtin_bit_cast(float, a[kPA[j]].x ^ b[t & 1][kPB[j]].x);
                continue;
            }
            acc[nb] = mma(a[kPA[j]], b[t & 1][kPB[j]], acc[nb]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// k steps [KS0, KS1) of a tile's product
template <int NKS, int NBLK, int KS0, int KS1>
PG_HD void tile_product(const TileA<NKS> &t, const char *image, int lane, f32x16 (&acc)[NBLK]) {
    if constexpr (kPipe) {
        f32x4 v[KS1 - KS0][2];
#pragma unroll
        for (int i = 0; i < KS1 - KS0; ++i) { v[i][0] = t.v[KS0 + i][0]; v[i][1] = t.v[KS0 + i][1]; }
        product_steps<NBLK, KS1 - KS0>(v, image, lane, KS0, acc);
        return;
    }
#pragma unroll
    for (int ks = KS0; ks < KS1; ++ks) {
        u32x4 a[3];
        split8(t.v[ks][0], t.v[ks][1], a);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            PGCN_DENSE_PRODUCTS;
            u32x4 b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(image + image_offset(pl, ks, nb, lane));
#pragma unroll
            for (int i = 0; i < 6; ++i)
                acc[nb] = mma(a[kPA[i]], b[kPB[i]], acc[nb]);
            __builtin_amdgcn_sched_barrier(0);     // (left alone the scheduler hoists every split and LDS read of the tile: spills)
        }
    }
}
template <int NBLK>
PG_HD void zero_acc(f32x16 (&acc)[NBLK]) {
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
}

template <int NBLK, int CNT>
PG_HD void half_product(const f32x4 (&v)[CNT][2], const char *image, int lane, int ks0, f32x16 (&acc)[NBLK]) {
    if constexpr (kPipe) {
        product_steps<NBLK, CNT>(v, image, lane, ks0, acc);
        return;
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        u32x4 a[3];
        split8(v[i][0], v[i][1], a);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            PGCN_DENSE_PRODUCTS;
            u32x4 b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(image + image_offset(pl, ks0 + i, nb, lane));
#pragma unroll
            for (int j = 0; j < 6; ++j)
                acc[nb] = mma(a[kPA[j]], b[kPB[j]], acc[nb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// C (n x N) = op(A) (n x K) . Bm (K x N); MASK: op(A) = A (.) [Y > 0] (and Gm = op(A) when Gm != nullptr); relu: C = relu(C).
template <int NKS, int NBLK, bool MASK>
__global__ __launch_bounds__(kThreads, 2) void dense_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ Y,
                                                            int64_t ldy, float *__restrict__ Gm, int64_t ldgm, int64_t n, int K,
                                                            int N, const float *__restrict__ W, int64_t ldw, int transposed,
                                                            float *__restrict__ C, int64_t ldc, int relu) {
    extern __shared__ __attribute__((aligned(16))) char image[];
    {
        constexpr int kMine = kSlotsPerPlane / kThreads;      // slots s = thread + 512 q: k step (thread >> 8) + 2 q, column block per wave
        float v[kMine][8];
#pragma unroll
        for (int q = 0; q < kMine; ++q) {
            const int s = (int)threadIdx.x + q * kThreads;
            if ((s >> 8) < NKS && ((s >> 6) & 3) < NBLK) slot_load(W, ldw, transposed, K, N, s, v[q]);
        }
#pragma unroll
        for (int q = 0; q < kMine; ++q) {
            const int s = (int)threadIdx.x + q * kThreads;
            if ((s >> 8) < NKS && ((s >> 6) & 3) < NBLK) slot_store(image, s, v[q]);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const int64_t stride = (int64_t)gridDim.x * kWaves;
    // first tile of this wave; kSpread: wave w of workgroup b = global wave w B + b, so that the tiles of the last, partial round
    // (7 281 tiles over 2 048 waves at the benchmark size) land on waves 0-3 (+ some 4) of EVERY CU instead of on all waves of the
    // first 142 workgroups
    int64_t tile = kSpread ? (int64_t)w * gridDim.x + blockIdx.x : (int64_t)blockIdx.x * kWaves + w;
    f32x16 acc[NBLK];
    if constexpr (MASK && kMaskPipe) {
        constexpr int H = NKS / 2;
        HalfRaw<H> raw;
        f32x4 v[H][2];
        if (tile < ntiles) load_half<H>(raw, A, lda, Y, ldy, tile * kRows, n, K, lane, 0);
        while (tile < ntiles) {
            const int64_t tn = tile + stride, row0 = tile * kRows;
            zero_acc(acc);
            mask_half<H>(v, raw, Gm, ldgm, row0, n, K, lane, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_half<H>(raw, A, lda, Y, ldy, row0, n, K, lane, H);                          // under the first half's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            half_product<NBLK, H>(v, image, lane, 0, acc);
            mask_half<H>(v, raw, Gm, ldgm, row0, n, K, lane, H);
            __builtin_amdgcn_sched_barrier(0);
            if (tn < ntiles) load_half<H>(raw, A, lda, Y, ldy, tn * kRows, n, K, lane, 0);    // under the second half's and the stores
            __builtin_amdgcn_sched_barrier(0);
            half_product<NBLK, H>(v, image, lane, H, acc);
            store_c(acc, NBLK, C, ldc, row0, n, N, lane, relu);
            tile = tn;
        }
    } else if constexpr (MASK) {
        for (; tile < ntiles; tile += stride) {
            TileA<NKS> cur;
            load_tile_masked<NKS>(cur, A, lda, Y, ldy, Gm, ldgm, tile * kRows, n, K, lane);
            zero_acc(acc);
            tile_product<NKS, NBLK, 0, NKS>(cur, image, lane, acc);
            store_c(acc, NBLK, C, ldc, tile * kRows, n, N, lane, relu);
        }
    } else if constexpr (kPipe) {
        // two register sets that swap roles: `from` is multiplied while `into` receives the wave's next tile
        TileA<NKS> t0, t1;
        auto one_tile = [&](const TileA<NKS> &from, TileA<NKS> &into) {
            const int64_t tn = tile + stride;
            zero_acc(acc);
            if constexpr (kProbe == 3) {              // timing only: no loads after the first tile
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) { asm volatile("" : "+v"(t0.v[ks][0]), "+v"(t0.v[ks][1])); }
                tile_product<NKS, NBLK, 0, NKS>(t0, image, lane, acc);
                store_c(acc, NBLK, C, ldc, tile * kRows, n, N, lane, relu);
                tile = tn;
                return;
            }
            if constexpr (kPrefetch == 2) {
                if (tn < ntiles) load_tile<NKS>(into, A, lda, tn * kRows, n, K, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_product<NKS, NBLK, 0, NKS / 2>(from, image, lane, acc);
            if constexpr (kPrefetch == 1) {
                if (tn < ntiles) load_tile<NKS>(into, A, lda, tn * kRows, n, K, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_product<NKS, NBLK, NKS / 2, NKS>(from, image, lane, acc);
            store_c(acc, NBLK, C, ldc, tile * kRows, n, N, lane, relu);
            if constexpr (kPrefetch == 0) {
                if (tn < ntiles) load_tile<NKS>(into, A, lda, tn * kRows, n, K, lane);
            }
            tile = tn;
        };
        if (tile < ntiles) load_tile<NKS>(t0, A, lda, tile * kRows, n, K, lane);
        while (tile < ntiles) {
            one_tile(t0, t1);
            if (tile >= ntiles) break;
            one_tile(t1, t0);
        }
    } else {
        TileA<NKS> cur, nxt;
        if (tile < ntiles) load_tile<NKS>(cur, A, lda, tile * kRows, n, K, lane);
        while (tile < ntiles) {
            const int64_t tn = tile + stride;
            zero_acc(acc);
            if constexpr (kPrefetch == 2) {           // probe: the whole next tile in flight from the start (spills)
                if (tn < ntiles) load_tile<NKS>(nxt, A, lda, tn * kRows, n, K, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_product<NKS, NBLK, 0, NKS / 2>(cur, image, lane, acc);
            // the next tile's loads go out once half of this one's registers are free, and land under the second half
            // of its MFMAs, its stores and the other wave of the SIMD
            if constexpr (kPrefetch == 1) {
                if (tn < ntiles) load_tile<NKS>(nxt, A, lda, tn * kRows, n, K, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            tile_product<NKS, NBLK, NKS / 2, NKS>(cur, image, lane, acc);
            store_c(acc, NBLK, C, ldc, tile * kRows, n, N, lane, relu);
            if constexpr (kPrefetch == 0) {           // probe: no prefetch (the other wave of the SIMD is the only overlap)
                if (tn < ntiles) load_tile<NKS>(nxt, A, lda, tn * kRows, n, K, lane);
            }
            cur = nxt;
            tile = tn;
        }
    }
}

template <int NKS, int NBLK, bool MASK>
int launch(const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K, int N,
           const float *W, int64_t ldw, int transposed, float *C, int64_t ldc, int relu, int workgroups, hipStream_t s) {
    auto kern = dense_kernel<NKS, NBLK, MASK>;
    static bool attr_set[64] = {false};
    static std::mutex attr_mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(-1, "hipGetDevice");
    {
        std::lock_guard<std::mutex> lock(attr_mu);          // (first calls from two threads)
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kImageBytes) != hipSuccess)
                return fail(-1, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)workgroups), dim3(kThreads), kImageBytes, s, A, lda, Y, ldy, Gm, ldgm, n, K, N, W, ldw,
                       transposed, C, ldc, relu);
    return hipGetLastError() == hipSuccess ? 0 : fail(-1, "kernel launch");
}

template <bool MASK>
int dispatch(const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K, int N,
             const float *W, int64_t ldw, int transposed, float *C, int64_t ldc, int relu, hipStream_t s) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return fail(-1, "hipDeviceGetAttribute(MultiprocessorCount)");
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const int64_t need = (ntiles + kWaves - 1) / kWaves;
    const int wgs = (int)(need < cus ? need : cus);           // one persistent workgroup per CU (96 KB of LDS each)
    const int nks = (K + 15) / 16, nblk = (N + 31) / 32;
#define PGCN_DENSE_CASE(KS, NB)                                                                                              \
    if (nks <= KS && nblk <= NB)                                                                                             \
        return launch<KS, NB, MASK>(A, lda, Y, ldy, Gm, ldgm, n, K, N, W, ldw, transposed, C, ldc, relu, wgs, s);
    PGCN_DENSE_CASE(4, 2)
    PGCN_DENSE_CASE(4, 4)
    PGCN_DENSE_CASE(8, 2)
    PGCN_DENSE_CASE(8, 4)
#undef PGCN_DENSE_CASE
    return fail(-2, "pgcn_dense: widths above 128");
}

}  // namespace pgcn_dense

#else  // PGCN_DENSE_HOST_EMU
// ---- host emulation: the same index functions, lane by lane, with the MFMA spelled out -----------------------------------
}  // namespace pgcn_dense

namespace {
using namespace pgcn_dense;

// D += A . B of v_mfma_f32_32x32x16_bf16 over the 64 lanes: A[m = lo][k = 8 hi + j] is element j of lane (lo, hi) of a,
// B[k = 8 hi + j][n = lo] element j of b, D[m = (r & 3) + 8 (r >> 2) + 4 hi][n = lo] register r of acc.
void mfma_emu(const u32x4 (&a)[64], const u32x4 (&b)[64], f32x16 (&acc)[64]) {
    float Am[32][16], Bm[16][32];
    for (int lane = 0; lane < 64; ++lane) {
        const int lo = lane & 31, hi = lane >> 5;
        for (int j = 0; j < 8; ++j) {
            const uint32_t wa = a[lane][j >> 1], wb = b[lane][j >> 1];
            Am[lo][8 * hi + j] = bf16_as_f32((j & 1) ? (wa >> 16) : (wa & 0xffffu));
            Bm[8 * hi + j][lo] = bf16_as_f32((j & 1) ? (wb >> 16) : (wb & 0xffffu));
        }
    }
    for (int lane = 0; lane < 64; ++lane) {
        const int lo = lane & 31, hi = lane >> 5;
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
            float s = acc[lane][r];
            for (int k = 0; k < 16; ++k) s += Am[m][k] * Bm[k][lo];
            acc[lane][r] = s;
        }
    }
}

// One launch with the kernel's template parameters: the image as the kernel fills it, every tile loaded by the kernel's own
// loaders (forward: load_tile; backward: load_half + mask_half, the two halves of a tile), stored by its store_c.
template <int NKS, int NBLK>
int emulate(int mode, const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K, int N,
            const float *W, int64_t ldw, float *C, int64_t ldc, int relu) {
    alignas(16) static char image[kImageBytes];
    memset(image, 0xff, sizeof image);                       // (slots the kernel does not fill must not be read)
    for (int s = 0; s < kSlotsPerPlane; ++s) {
        const int ks = s >> 8, nb = (s >> 6) & 3;
        if (ks < NKS && nb < NBLK) {
            float v[8];
            slot_load(W, ldw, mode == 0, K, N, s, v);
            slot_store(image, s, v);
        }
    }
    const int64_t ntiles = (n + kRows - 1) / kRows;
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        static TileA<NKS> t[64];
        for (int lane = 0; lane < 64; ++lane) {
            if (mode == 0) {
                load_tile<NKS>(t[lane], A, lda, tile * kRows, n, K, lane);
            } else {
                constexpr int H = NKS / 2;
                for (int half = 0; half < 2; ++half) {
                    HalfRaw<H> raw;
                    f32x4 v[H][2];
                    load_half<H>(raw, A, lda, Y, ldy, tile * kRows, n, K, lane, half * H);
                    mask_half<H>(v, raw, Gm, ldgm, tile * kRows, n, K, lane, half * H);
                    for (int i = 0; i < H; ++i) { t[lane].v[half * H + i][0] = v[i][0]; t[lane].v[half * H + i][1] = v[i][1]; }
                }
            }
        }
        static f32x16 acc[NBLK][64];
        for (int nb = 0; nb < NBLK; ++nb)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 16; ++r) acc[nb][lane][r] = 0.f;
        for (int ks = 0; ks < NKS; ++ks) {
            static u32x4 a[3][64], b[3][64];
            for (int lane = 0; lane < 64; ++lane) {
                u32x4 p[3];
                split8(t[lane].v[ks][0], t[lane].v[ks][1], p);
                for (int pl = 0; pl < 3; ++pl) a[pl][lane] = p[pl];
            }
            for (int nb = 0; nb < NBLK; ++nb) {
                for (int lane = 0; lane < 64; ++lane)
                    for (int pl = 0; pl < 3; ++pl) memcpy(&b[pl][lane], image + image_offset(pl, ks, nb, lane), 16);
                PGCN_DENSE_PRODUCTS;
                for (int i = 0; i < 6; ++i) {
                    if (kCT) mfma_emu(b[kPB[i]], a[kPA[i]], acc[nb]);      // (the kernel's mma(): operands swapped)
                    else mfma_emu(a[kPA[i]], b[kPB[i]], acc[nb]);
                }
            }
        }
        for (int lane = 0; lane < 64; ++lane) {
            f32x16 mine[NBLK];
            for (int nb = 0; nb < NBLK; ++nb) mine[nb] = acc[nb][lane];
            store_c(mine, NBLK, C, ldc, tile * kRows, n, N, lane, relu);
        }
    }
    return 0;
}
}  // namespace

// mode 0: C = [relu](A . W^T), W: N x K;  mode 1: Gm = A (.) [Y > 0] (when Gm), C = Gm . W, W: K x N.
extern "C" int pgcn_dense_emulate(int mode, const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm,
                                  int64_t n, int K, int N, const float *W, int64_t ldw, float *C, int64_t ldc, int relu);
namespace pgcn_dense {
typedef void *hipStream_t;
// the host build's stand-in for the launch: HOST pointers, the stream is ignored
template <bool MASK>
int dispatch(const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K, int N,
             const float *W, int64_t ldw, int transposed, float *C, int64_t ldc, int relu, hipStream_t) {
    if (transposed != (MASK ? 0 : 1)) return fail(-1, "emulation: mode / layout mismatch");
    return pgcn_dense_emulate(MASK ? 1 : 0, A, lda, Y, ldy, Gm, ldgm, n, K, N, W, ldw, C, ldc, relu);
}
}  // namespace pgcn_dense
extern "C" int pgcn_dense_emulate(int mode, const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm,
                                  int64_t n, int K, int N, const float *W, int64_t ldw, float *C, int64_t ldc, int relu) {
    if (K <= 0 || N <= 0 || K > kMaxF || N > kMaxF || K % 4) return -2;
    const int nks = (K + 15) / 16, nblk = (N + 31) / 32;      // the kernel's own choice of instantiation (dispatch)
#define PGCN_DENSE_CASE(KS, NB) \
    if (nks <= KS && nblk <= NB) return emulate<KS, NB>(mode, A, lda, Y, ldy, Gm, ldgm, n, K, N, W, ldw, C, ldc, relu);
    PGCN_DENSE_CASE(4, 2)
    PGCN_DENSE_CASE(4, 4)
    PGCN_DENSE_CASE(8, 2)
    PGCN_DENSE_CASE(8, 4)
#undef PGCN_DENSE_CASE
    return -2;
}
#endif

// ---- the C ABI (include/pgcn_gemm.h); in the host build the same checks in front of the emulator ------------------------------
extern "C" const char *pgcn_dense_last_error(void) { return pgcn_dense::g_err; }

// Y (n x fout, ldy) = [relu] (X (n x fin, ldx) . W^T),  W: fout x fin row-major (nn.Linear's weight), on `stream`.
// 0; -2: shape / alignment outside what the kernel takes (the caller uses the library GEMM); -1: errors.
extern "C" int pgcn_linear_relu_f32(const float *X, int64_t ldx, int64_t n, int32_t fin, const float *W, int64_t ldw, int32_t fout,
                                    float *Y, int64_t ldy, int32_t relu, void *stream) {
    using namespace pgcn_dense;
    if (int rc = check(X, ldx, n, fin, fout, W, ldw, fout, fin, Y, ldy)) return rc;
    if (n == 0) return 0;
    return dispatch<false>(X, ldx, nullptr, 0, nullptr, 0, n, fin, fout, W, ldw, 1, Y, ldy, relu ? 1 : 0, (hipStream_t)stream);
}

// Gm = G (.) [Y > 0] (written when Gm != NULL; may be G itself),  dX (n x fin, lddx) = Gm . W;  G, Y, Gm: n x fout.
extern "C" int pgcn_linear_relu_grad_input_f32(const float *G, int64_t ldg, const float *Y, int64_t ldy, float *Gm, int64_t ldgm,
                                               int64_t n, int32_t fout, const float *W, int64_t ldw, int32_t fin, float *dX,
                                               int64_t lddx, void *stream) {
    using namespace pgcn_dense;
    if (int rc = check(G, ldg, n, fout, fin, W, ldw, fout, fin, dX, lddx)) return rc;
    if (!Y && n > 0) return fail(-1, "pgcn_linear_relu_grad_input_f32: Y is NULL");
    if (ldy % 4 || (uintptr_t)Y % 16 || ldy < fout) return fail(-2, "pgcn_dense: rows of Y must be 16-byte pieces");
    if (Gm && (ldgm % 4 || (uintptr_t)Gm % 16 || ldgm < fout)) return fail(-2, "pgcn_dense: rows of Gm must be 16-byte pieces");
    if (n == 0) return 0;
    return dispatch<true>(G, ldg, Y, ldy, Gm, ldgm, n, fout, fin, W, ldw, 0, dX, lddx, 0, (hipStream_t)stream);
}
