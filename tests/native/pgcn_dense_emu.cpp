// pgcn_dense_emu.cpp -- TEST INFRASTRUCTURE: a host build of the index arithmetic of gemm/pgcn_dense.hip.
// The kernel's own functions (gemm/pgcn_dense_tile.h: LDS image slots of W, a lane's 16-byte pieces of a tile, the accumulator
// layout, the argument checks) are run lane by lane around an emulated v_mfma_f32_32x32x16_bf16 whose operand layout is the one
// csrc/pgcn_spmm_dense3.hip runs on hardware; the same C entry points as the library (include/pgcn_gemm.h) on HOST pointers, the
// stream argument ignored.  tests/test_zz_dense_fused.py compiles this file with clang++ and binds it like the library.
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
                for (int i = 0; i < 6; ++i) mfma_emu(a[kPA[i]], b[kPB[i]], acc[nb]);
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
static int emulate_any(int mode, const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K,
                       int N, const float *W, int64_t ldw, float *C, int64_t ldc, int relu) {
    if (K <= 0 || N <= 0 || K > kMaxF || N > kMaxF || K % 4) return -2;
    const int nks = (K + 15) / 16, nblk = (N + 31) / 32;      // the kernel's own choice of instantiation (its dispatch())
#define PGCN_DENSE_CASE(KS, NB) \
    if (nks <= KS && nblk <= NB) return emulate<KS, NB>(mode, A, lda, Y, ldy, Gm, ldgm, n, K, N, W, ldw, C, ldc, relu);
    PGCN_DENSE_CASE(4, 2)
    PGCN_DENSE_CASE(4, 4)
    PGCN_DENSE_CASE(8, 2)
    PGCN_DENSE_CASE(8, 4)
#undef PGCN_DENSE_CASE
    return -2;
}

// ---- the C ABI of include/pgcn_gemm.h: the library's checks in front of the emulator ------------------------------------------
extern "C" const char *pgcn_dense_last_error(void) { return pgcn_dense::g_err; }

extern "C" int pgcn_linear_relu_f32(const float *X, int64_t ldx, int64_t n, int32_t fin, const float *W, int64_t ldw, int32_t fout,
                                    float *Y, int64_t ldy, int32_t relu, void *) {
    if (int rc = check(X, ldx, n, fin, fout, W, ldw, fout, fin, Y, ldy)) return rc;
    if (n == 0) return 0;
    return emulate_any(0, X, ldx, nullptr, 0, nullptr, 0, n, fin, fout, W, ldw, Y, ldy, relu ? 1 : 0);
}

extern "C" int pgcn_linear_relu_grad_input_f32(const float *G, int64_t ldg, const float *Y, int64_t ldy, float *Gm, int64_t ldgm,
                                               int64_t n, int32_t fout, const float *W, int64_t ldw, int32_t fin, float *dX,
                                               int64_t lddx, void *) {
    if (int rc = check(G, ldg, n, fout, fin, W, ldw, fout, fin, dX, lddx)) return rc;
    if (!Y && n > 0) return fail(-1, "pgcn_linear_relu_grad_input_f32: Y is NULL");
    if (ldy % 4 || (uintptr_t)Y % 16 || ldy < fout) return fail(-2, "pgcn_dense: rows of Y must be 16-byte pieces");
    if (Gm && (ldgm % 4 || (uintptr_t)Gm % 16 || ldgm < fout)) return fail(-2, "pgcn_dense: rows of Gm must be 16-byte pieces");
    if (n == 0) return 0;
    return emulate_any(1, G, ldg, Y, ldy, Gm, ldgm, n, fout, fin, W, ldw, dX, lddx, 0);
}
