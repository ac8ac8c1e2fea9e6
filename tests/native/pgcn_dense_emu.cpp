// pgcn_dense_emu.cpp -- TEST INFRASTRUCTURE: a host build of the index arithmetic of gemm/pgcn_dense.hip.
// The kernel's own functions (gemm/pgcn_dense_tile.h: LDS image slots of W, a lane's 16-byte pieces of a tile and their window
// offsets, the accumulator layout, the sign-mask words, the argument checks) are run lane by lane around an emulated
// v_mfma_f32_32x32x16_bf16 whose operand layout is the one csrc/pgcn_spmm_dense3.hip runs on hardware; the same C entry points as
// the library (include/pgcn_gemm.h) on HOST pointers, the stream argument ignored.  tests/test_zz_dense_fused.py compiles this file
// with clang++ and binds it like the library.  (The schedule of the device kernel -- what is in flight when -- is not emulated: only
// which bytes a lane reads and writes and what it multiplies.)
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define PGCN_DENSE_HOST_EMU 1
#define PG_HD inline

namespace pgcn_dense {
#include "pgcn_dense_tile.h"
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

struct EmuArgs {
    const float *A;
    int64_t lda;
    const uint32_t *mask_in;
    float *Gm;
    int64_t ldgm;
    uint32_t *mask_out;
    int64_t n;
    int K, N;
    const float *W;
    int64_t ldw;
    int transposed;
    float *C;
    int64_t ldc;
    int relu;
};

// One launch with the kernel's template parameters: the image as the kernel fills it, every tile through the kernel's own windows,
// loaders, mask arithmetic and stores.  MODE 0: forward; 1: input gradient.
template <int NKS, int NBLK, int MODE, bool RAGGED>
int emulate(const EmuArgs &a) {
    alignas(16) static char image[kImageBytes];
    memset(image, 0xff, sizeof image);                       // (slots the kernel does not fill must not be read)
    const bool vec = slot_vec_ok(a.W, a.ldw, a.transposed, a.K);
    for (int s = 0; s < kSlotsPerPlane; ++s) {
        const int ks = s >> 8, nb = (s >> 6) & 3;
        if (ks < NKS && nb < NBLK) {
            float v[8];
            slot_load(a.W, a.ldw, a.transposed, a.K, a.N, s, v, vec);
            slot_store(image, s, v);
        }
    }
    const int64_t n = a.n, ntiles = (n + kRows - 1) / kRows;
    const int K = a.K, N = a.N, mwK = mask_words(K), mwN = mask_words(N);
    auto win_of = [&](const void *M, int64_t ld, int64_t t, int width) { return tile_window(M, ld, M ? n : 0, t * kRows, width); };
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        static u32x4 ap[NKS][3][64];
        const window_t wa = win_of(a.A, a.lda, tile, K), wgm = win_of(a.Gm, a.ldgm, tile, K), wc = win_of(a.C, a.ldc, tile, N);
        const window_t wmi = win_of(a.mask_in, mwK, tile, mwK), wmo = win_of(a.mask_out, mwN, tile, mwN);
        for (int lane = 0; lane < 64; ++lane) {
            const uint32_t a_off = piece_lane_offset(a.lda, lane), gm_off = piece_lane_offset(a.ldgm, lane);
            uint32_t mwn[4];
            const uint32_t no_mask = a.mask_in ? 0u : ~0u;
            for (int q = 0; q < 4; ++q)
                mwn[q] = (q < mwK ? win_load4u(wmi, (uint32_t)((lane & 31) * mwK * 4) + 4 * (q < mwK ? q : 0), 0) : 0u) | no_mask;
            for (int ks = 0; ks < NKS; ++ks) {
                f32x4 raw[2];
                load_piece<RAGGED>(raw, wa, a_off, ks, K, lane);
                if (MODE == 1) {
                    for (int h = 0; h < 2; ++h) raw[h] = mask4_bits(raw[h], mwn[ks >> 1], ks, h, lane);
                    store_piece<RAGGED>(raw, wgm, gm_off, ks, K, lane);
                }
                u32x4 p[3];
                split8(raw[0], raw[1], p);
                for (int pl = 0; pl < 3; ++pl) ap[ks][pl][lane] = p[pl];
            }
        }
        for (int nb = 0; nb < NBLK; ++nb) {
            static f32x16 acc[64];
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 16; ++r) acc[lane][r] = 0.f;
            for (int ks = 0; ks < NKS; ++ks) {
                static u32x4 b[3][64];
                for (int lane = 0; lane < 64; ++lane)
                    for (int pl = 0; pl < 3; ++pl) memcpy(&b[pl][lane], image + image_offset(pl, ks, nb, lane), 16);
                PGCN_DENSE_PRODUCTS;
                for (int i = 0; i < 6; ++i) mfma_emu(ap[ks][kPA[i]], b[kPB[i]], acc);
            }
            for (int lane = 0; lane < 64; ++lane)
                store_regs<RAGGED>(acc[lane], 0, 16, nb, wc, acc_lane_offset(a.ldc, lane), (uint32_t)(a.ldc * 4), N, lane, a.relu);
            if (MODE == 0) {
                // the device: ballot of (register r > 0) over the wave, word of (r, h) collected in lane 16 h + r, lanes < 32 store
                uint32_t mword[64] = {0};
                for (int r = 0; r < 16; ++r) {
                    uint64_t bal = 0;
                    for (int lane = 0; lane < 64; ++lane) bal |= (uint64_t)(mask_bit_of(acc[lane][r]) ? 1 : 0) << lane;
                    mword[r] = (uint32_t)bal;
                    mword[16 + r] = (uint32_t)(bal >> 32);
                }
                for (int lane = 0; lane < 32; ++lane)
                    if (nb < mwN) win_store4u(wmo, (uint32_t)(mask_row_of_lane(lane) * mwN * 4) + 4 * nb, 0, mword[lane]);
            }
        }
    }
    return 0;
}

template <int MODE>
int emulate_any(const EmuArgs &a) {
    if (a.K <= 0 || a.N <= 0 || a.K > kMaxF || a.N > kMaxF || a.K % 4) return -2;
    const int nks = (a.K + 15) / 16, nblk = (a.N + 31) / 32;      // the kernel's own choice of instantiation (its dispatch())
#define PGCN_DENSE_CASE(KS, NB)                                                    \
    if (nks <= KS && nblk <= NB) {                                                 \
        if (a.K == 16 * KS && a.N == 32 * NB) return emulate<KS, NB, MODE, false>(a); \
        return emulate<KS, NB, MODE, true>(a);                                     \
    }
    PGCN_DENSE_CASE(4, 2)
    PGCN_DENSE_CASE(4, 4)
    PGCN_DENSE_CASE(8, 2)
    PGCN_DENSE_CASE(8, 4)
#undef PGCN_DENSE_CASE
    return -2;
}
}  // namespace

// ---- the C ABI of include/pgcn_gemm.h: the library's checks in front of the emulator ------------------------------------------
extern "C" const char *pgcn_dense_last_error(void) { return pgcn_dense::g_err; }

extern "C" int pgcn_linear_relu_f32(const float *X, int64_t ldx, int64_t n, int32_t fin, const float *W, int64_t ldw, int32_t fout,
                                    float *Y, int64_t ldy, int32_t relu, uint32_t *mask, void *) {
    if (int rc = check(X, ldx, n, fin, fout, W, ldw, fout, fin, Y, ldy)) return rc;
    if (n == 0) return 0;
    EmuArgs a{};
    a.A = X; a.lda = ldx; a.mask_out = mask; a.n = n; a.K = fin; a.N = fout; a.W = W; a.ldw = ldw; a.transposed = 1; a.C = Y; a.ldc = ldy;
    a.relu = relu ? 1 : 0;
    return emulate_any<0>(a);
}

extern "C" int pgcn_linear_relu_grad_input_f32(const float *G, int64_t ldg, const uint32_t *mask, float *Gm, int64_t ldgm, int64_t n,
                                               int32_t fout, const float *W, int64_t ldw, int32_t fin, float *dX, int64_t lddx, void *) {
    if (int rc = check(G, ldg, n, fout, fin, W, ldw, fout, fin, dX, lddx)) return rc;
    if (Gm && (ldgm % 4 || (uintptr_t)Gm % 16 || ldgm < fout)) return fail(-2, "pgcn_dense: rows of Gm must be 16-byte pieces");
    if (n == 0) return 0;
    EmuArgs a{};
    a.A = G; a.lda = ldg; a.mask_in = mask; a.Gm = Gm; a.ldgm = ldgm; a.n = n; a.K = fout; a.N = fin; a.W = W; a.ldw = ldw; a.transposed = 0;
    a.C = dX; a.ldc = lddx; a.relu = 0;
    return emulate_any<1>(a);
}

extern "C" int pgcn_sign_mask_f32(const float *Y, int64_t ldy, int64_t n, int32_t N, uint32_t *mask, void *) {
    if (n < 0 || N <= 0 || ldy < N || (n > 0 && (!Y || !mask))) return fail(-1, "pgcn_sign_mask_f32: bad argument");
    const int mw = (N + 31) / 32;
    for (int64_t row = 0; row < n; ++row)
        for (int w = 0; w < mw; ++w) {
            uint32_t bits = 0;
            for (int b = 0; b < 32; ++b) {
                const int c = 32 * w + b;
                if (c < N && mask_bit_of(Y[row * ldy + c])) bits |= 1u << b;
            }
            mask[row * mw + w] = bits;
        }
    return 0;
}
