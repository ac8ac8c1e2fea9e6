// pgcn_dense_tile.h -- the index arithmetic of gemm/pgcn_dense.hip: where a value of W lands in the LDS image of B operands,
// which 16-byte pieces of a tile a lane loads, which element of C an accumulator register is, the argument checks.  Shared by
//   * gemm/pgcn_dense.hip (device build: PG_HD = __device__ __forceinline__), and
//   * tests/native/pgcn_dense_emu.cpp (host build, -DPGCN_DENSE_HOST_EMU: the same functions run lane by lane around an
//     emulated v_mfma_f32_32x32x16_bf16, so that slots, operand lanes and the accumulator layout are checked on the CPU).
// Included INSIDE namespace pgcn_dense; PG_HD comes from the including file.
//
// Loads and stores of tiles that lie inside the matrices (all but a wave's last tile of a full-width operand) take an
// unpredicated path off ONE address (r05, tools/probes_r05: the guarded pieces cost ~12 instructions with three quarter-rate
// 64-bit multiplies per 4-byte store; forward 75.7 -> 65.3 us at n = 232 965, f = 128), stores of C carry the non-temporal hint
// (C is not read again by this kernel: 65.3 -> 63.3 us, input gradient 119.9 -> 116.8 us).
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
// dropped afterwards).
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

PG_HD void store1(float *p, float x) {
#ifndef PGCN_DENSE_HOST_EMU
    __builtin_nontemporal_store(x, p);
#else
    *p = x;
#endif
}
// accumulator register r of lane (lo, hi), column block nb -> element (row0 + (r & 3) + 8 (r >> 2) + 4 hi, 32 nb + lo) of C
PG_HD void store_c(const f32x16 *acc, int nblk, float *C, int64_t ldc, int64_t row0, int64_t n, int N, int lane, int relu) {
    const int hi = lane >> 5, lo = lane & 31;
    if (row0 + kRows <= n && 32 * nblk == N) {        // (wave-uniform) the tile lies inside C: no predicates
        float *base = C + (row0 + 4 * hi) * ldc + lo;
#pragma unroll
        for (int nb = 0; nb < nblk; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                store1(base + (int64_t)((r & 3) + 8 * (r >> 2)) * ldc + 32 * nb, relu ? relu1(acc[nb][r]) : acc[nb][r]);
        return;
    }
#pragma unroll
    for (int nb = 0; nb < nblk; ++nb) {
        const int col = 32 * nb + lo;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < n) store1(&C[row * ldc + col], relu ? relu1(acc[nb][r]) : acc[nb][r]);
        }
    }
}

// ---- argument checks and the error string ---------------------------------------------------------------------------------
inline thread_local char g_err[256] = "";
inline int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

inline int check(const void *A, int64_t lda, int64_t n, int K, int N, const void *W, int64_t ldw, int wrows, int wcols,
                 const void *C, int64_t ldc) {
    if (n < 0 || K <= 0 || N <= 0 || !W || (n > 0 && (!A || !C))) return fail(-1, "pgcn_dense: bad argument");
    if (K > kMaxF || N > kMaxF) return fail(-2, "pgcn_dense: widths above 128 are left to the library GEMM");
    if (K % 4 || lda % 4 || (uintptr_t)A % 16) return fail(-2, "pgcn_dense: rows of the left operand must be 16-byte pieces");
    if (lda < K || ldc < N || ldw < wcols || wrows <= 0) return fail(-1, "pgcn_dense: leading dimension below the width");
    if (n > ((int64_t)1 << 40)) return fail(-1, "pgcn_dense: n out of range");
    return 0;
}

// ---- a wave's operand tiles -----------------------------------------------------------------------------------------------------
template <int NKS>
struct TileA {
    f32x4 v[NKS][2];
};

template <int NKS>
PG_HD void load_tile(TileA<NKS> &t, const float *__restrict__ A, int64_t lda, int64_t row0, int64_t n, int K, int lane) {
    if (row0 + kRows <= n && 16 * NKS == K) {         // (wave-uniform) the tile lies inside A: 16 loads off one address
        const float *base = A + (row0 + (lane & 31)) * lda + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) t.v[ks][h] = *reinterpret_cast<const f32x4 *>(base + 16 * ks + 4 * h);
        return;
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

// ---- the masked operand as a stream of HALF tiles (k steps [ks0, ks0 + CNT) of a tile) ----------------------------------------
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
    if (row0 + kRows <= n && 32 * CNT == K) {         // (wave-uniform) inside the matrix: no predicates
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
// G (.) [Y > 0] of a half, written out as Gm (when asked for) on the way
template <int CNT>
PG_HD void mask_half(f32x4 (&v)[CNT][2], const HalfRaw<CNT> &r, float *__restrict__ Gm, int64_t ldgm, int64_t row0, int64_t n, int K,
                     int lane, int ks0) {
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

// ---- the operand as the SUM of a row's partial rows: the fix-up of the aggregation folded into its consumer (r05) ---------------
// The producers of one aggregation (gather tasks, strips, bf16 blocks) leave partial rows in a work-space; csrc's
// spmm_fixup_list_kernel adds a row's partial rows in list order and writes A.H, which the dense kernel then reads back.  Here
// the dense kernel's loader does that sum itself, in the same order (bit-identical operand), so A.H is never written:
//     row_fix[r] = {begin, count};  count >= 0:  S[r] = ((0 + P[id_0]) + P[id_1]) + ...,  id_t = slot_ids[begin + t]
//                                                (id_t = begin + t when slot_ids == NULL),  P[i] = partial + i * ldp
//                                   count <  0:  S[r] = base[r]   (a row some producer wrote directly)
// A lane's 16-byte pieces are those of piece_of(); the ids of up to kIdChunk slots are fetched together, the pieces of slot
// t + 1 are in flight while those of slot t are added.  Loads are unconditional (clamped ids, the sum kept by a select): a
// branch per slot would fence the loads in flight.  `tmax` >= every lane's count (wave-uniform on the device).
struct RowFix {
    int32_t begin, count;
};
constexpr int kIdChunk = 8;

template <int CNT>
PG_HD void load_row_pieces(f32x4 (&x)[CNT][2], const float *__restrict__ rowp, int K, int lane, int ks0, bool full) {
    // rowp = first element of the lane's row; pieces (ks0 + i, h) at 16 (ks0 + i) + 8 hi + 4 h
    const float *p = rowp + 8 * (lane >> 5) + 16 * ks0;
    if (full) {
#pragma unroll
        for (int i = 0; i < CNT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) x[i][h] = *reinterpret_cast<const f32x4 *>(p + 16 * i + 4 * h);
        return;
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 16 * (ks0 + i) + 8 * (lane >> 5) + 4 * h;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            x[i][h] = k < K ? *reinterpret_cast<const f32x4 *>(p + 16 * i + 4 * h) : z;
        }
}

template <int CNT>
PG_HD void sum_half(f32x4 (&v)[CNT][2], const float *__restrict__ partial, int64_t ldp, const int32_t *__restrict__ slot_ids,
                    const float *__restrict__ base, int64_t ldbase, int64_t row, RowFix rf, int tmax, int K, int lane, int ks0) {
    const bool full = 16 * (ks0 + CNT) <= K;          // (uniform) every piece of this half lies inside the width
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CNT; ++i) { v[i][0] = z; v[i][1] = z; }
    if (rf.count < 0) load_row_pieces<CNT>(v, base + row * ldbase, K, lane, ks0, full);     // (rare: a branch)
    for (int t0 = 0; t0 < tmax; t0 += kIdChunk) {
        int32_t id[kIdChunk];
#pragma unroll
        for (int j = 0; j < kIdChunk; ++j) {
            const int t = t0 + j;
            const bool on = t < rf.count;
            const int32_t at = on ? rf.begin + t : 0;
            const int32_t got = slot_ids ? slot_ids[at] : at;
            id[j] = on ? got : -1;
        }
        f32x4 x[2][CNT][2];
        load_row_pieces<CNT>(x[0], partial + (int64_t)(id[0] < 0 ? 0 : id[0]) * ldp, K, lane, ks0, full);
#pragma unroll
        for (int j = 0; j < kIdChunk; ++j) {
            if (t0 + j >= tmax) break;                // (uniform)
            if (j + 1 < kIdChunk && t0 + j + 1 < tmax)
                load_row_pieces<CNT>(x[(j + 1) & 1], partial + (int64_t)(id[j + 1] < 0 ? 0 : id[j + 1]) * ldp, K, lane, ks0, full);
            const bool on = id[j] >= 0;
#pragma unroll
            for (int i = 0; i < CNT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x4 s = v[i][h] + x[j & 1][i][h];
                    v[i][h].x = on ? s.x : v[i][h].x; v[i][h].y = on ? s.y : v[i][h].y;
                    v[i][h].z = on ? s.z : v[i][h].z; v[i][h].w = on ? s.w : v[i][h].w;
                }
        }
    }
}
// the summed half written out (the operand of the weight gradient in the backward; the aggregation itself when a caller wants it)
template <int CNT>
PG_HD void store_half(const f32x4 (&v)[CNT][2], float *__restrict__ S, int64_t lds, int64_t row0, int64_t n, int K, int lane, int ks0) {
    if (row0 + kRows <= n && 16 * (ks0 + CNT) <= K) {
        float *sb = S + (row0 + (lane & 31)) * lds + 8 * (lane >> 5) + 16 * ks0;
#pragma unroll
        for (int i = 0; i < CNT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4 *>(sb + 16 * i + 4 * h) = v[i][h];
        return;
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const Piece pm = piece_of(row0, n, K, lds, lane, ks0 + i, h);
            if (pm.ok) *reinterpret_cast<f32x4 *>(S + pm.off) = v[i][h];
        }
}
// C = the product where M > 0, else 0 (threshold_backward by the layer input M = relu(...) of the layer below: its mask pass
// folded into this layer's input gradient); accumulator layout as in store_c
PG_HD void store_c_masked(const f32x16 *acc, int nblk, float *C, int64_t ldc, const float *__restrict__ M, int64_t ldm, int64_t row0,
                          int64_t n, int N, int lane) {
    const int hi = lane >> 5, lo = lane & 31;
    if (row0 + kRows <= n && 32 * nblk == N) {
        float *base = C + (row0 + 4 * hi) * ldc + lo;
        const float *mb = M + (row0 + 4 * hi) * ldm + lo;
#pragma unroll
        for (int nb = 0; nb < nblk; ++nb) {
            float m[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = mb[(int64_t)((r & 3) + 8 * (r >> 2)) * ldm + 32 * nb];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                store1(base + (int64_t)((r & 3) + 8 * (r >> 2)) * ldc + 32 * nb, m[r] <= 0.f ? 0.f : acc[nb][r]);
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < nblk; ++nb) {
        const int col = 32 * nb + lo;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < n) store1(&C[row * ldc + col], M[row * ldm + col] <= 0.f ? 0.f : acc[nb][r]);
        }
    }
}

inline int check_fixup(const void *row_fix, const void *partial, int64_t ldp, const void *base, int64_t ldbase, int64_t n, int K, int N,
                       const void *W, int64_t ldw, int wcols, const void *S, int64_t lds, const void *M, int64_t ldm, const void *C,
                       int64_t ldc, int epilogue) {
    if (n < 0 || K <= 0 || N <= 0 || !W || (n > 0 && (!row_fix || !C)) || epilogue < 0 || epilogue > 2)
        return fail(-1, "pgcn_fixup_linear_f32: bad argument");
    if (K > kMaxF || N > kMaxF) return fail(-2, "pgcn_fixup_linear_f32: widths above 128 are left to the separate fix-up + library GEMM");
    if (K % 4) return fail(-2, "pgcn_fixup_linear_f32: the width of the summed rows must be a multiple of 4");
    if (partial && (ldp % 4 || (uintptr_t)partial % 16 || ldp < K)) return fail(-2, "pgcn_fixup_linear_f32: partial rows must be 16-byte pieces");
    if (base && (ldbase % 4 || (uintptr_t)base % 16 || ldbase < K)) return fail(-2, "pgcn_fixup_linear_f32: base rows must be 16-byte pieces");
    if (S && (lds % 4 || (uintptr_t)S % 16 || lds < K)) return fail(-2, "pgcn_fixup_linear_f32: rows of S must be 16-byte pieces");
    if (epilogue == 2 && (!M || ldm < N)) return fail(-1, "pgcn_fixup_linear_f32: the mask epilogue needs M");
    if (ldc < N || ldw < wcols) return fail(-1, "pgcn_fixup_linear_f32: leading dimension below the width");
    return 0;
}
