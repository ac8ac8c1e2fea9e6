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
// mode 2 (the fix-up folded into the product): the operand tiles come from sum_half, the epilogue may be the mask by M
struct FixupArgs {
    const RowFix *row_fix = nullptr;
    const int32_t *slot_ids = nullptr;
    const float *partial = nullptr;
    int64_t ldp = 0;
    const float *base = nullptr;
    int64_t ldbase = 0;
    float *S_out = nullptr;
    int64_t lds = 0;
    const float *M = nullptr;
    int64_t ldm = 0;
    int transposed = 1, epilogue = 0;
};

template <int NKS, int NBLK>
int emulate(int mode, const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K, int N,
            const float *W, int64_t ldw, float *C, int64_t ldc, int relu, const FixupArgs *fx = nullptr) {
    alignas(16) static char image[kImageBytes];
    memset(image, 0xff, sizeof image);                       // (slots the kernel does not fill must not be read)
    for (int s = 0; s < kSlotsPerPlane; ++s) {
        const int ks = s >> 8, nb = (s >> 6) & 3;
        if (ks < NKS && nb < NBLK) {
            float v[8];
            slot_load(W, ldw, mode == 2 ? fx->transposed : mode == 0, K, N, s, v);
            slot_store(image, s, v);
        }
    }
    const int64_t ntiles = (n + kRows - 1) / kRows;
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        static TileA<NKS> t[64];
        for (int lane = 0; lane < 64; ++lane) {
            if (mode == 0) {
                load_tile<NKS>(t[lane], A, lda, tile * kRows, n, K, lane);
            } else if (mode == 2) {
                constexpr int H = NKS / 2;
                const int64_t row = tile * kRows + (lane & 31);
                RowFix rf = {0, 0};
                if (row < n) rf = fx->row_fix[row];
                const int tmax = fx->partial ? (rf.count > 0 ? rf.count : 0) : 0;       // (the kernel: the wave's maximum)
                for (int half = 0; half < 2; ++half) {
                    f32x4 v[H][2];
                    sum_half<H>(v, fx->partial, fx->ldp, fx->slot_ids, fx->base, fx->ldbase, row < n ? row : 0, rf, tmax, K, lane, half * H);
                    if (fx->S_out) store_half<H>(v, fx->S_out, fx->lds, tile * kRows, n, K, lane, half * H);
                    for (int i = 0; i < H; ++i) { t[lane].v[half * H + i][0] = v[i][0]; t[lane].v[half * H + i][1] = v[i][1]; }
                }
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
            if (mode == 2 && fx->epilogue == 2) store_c_masked(mine, NBLK, C, ldc, fx->M, fx->ldm, tile * kRows, n, N, lane);
            else store_c(mine, NBLK, C, ldc, tile * kRows, n, N, lane, mode == 2 ? fx->epilogue : relu);
        }
    }
    return 0;
}
}  // namespace

// mode 0: C = [relu](A . W^T), W: N x K;  mode 1: Gm = A (.) [Y > 0] (when Gm), C = Gm . W, W: K x N.
static int emulate_any(int mode, const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K,
                       int N, const float *W, int64_t ldw, float *C, int64_t ldc, int relu, const FixupArgs *fx = nullptr) {
    if (K <= 0 || N <= 0 || K > kMaxF || N > kMaxF || K % 4) return -2;
    const int nks = (K + 15) / 16, nblk = (N + 31) / 32;      // the kernel's own choice of instantiation (its dispatch())
#define PGCN_DENSE_CASE(KS, NB) \
    if (nks <= KS && nblk <= NB) return emulate<KS, NB>(mode, A, lda, Y, ldy, Gm, ldgm, n, K, N, W, ldw, C, ldc, relu, fx);
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

extern "C" int pgcn_fixup_linear_f32(const int32_t *row_fix, const int32_t *slot_ids, const float *partial, int64_t ldp,
                                     const float *base, int64_t ldbase, int64_t n, int32_t k, const float *W, int64_t ldw,
                                     int32_t wrows, int32_t wcols, int32_t transposed, float *S_out, int64_t lds, const float *M,
                                     int64_t ldm, float *C, int64_t ldc, int32_t epilogue, void *) {
    if (wrows <= 0 || wcols <= 0 || (transposed ? wcols : wrows) != k) return fail(-1, "pgcn_fixup_linear_f32: W does not match the width of S");
    const int N = transposed ? wrows : wcols;
    if (int rc = check_fixup(row_fix, partial, ldp, base, ldbase, n, k, N, W, ldw, wcols, S_out, lds, M, ldm, C, ldc, epilogue)) return rc;
    if (n == 0) return 0;
    FixupArgs fx;
    fx.row_fix = reinterpret_cast<const RowFix *>(row_fix); fx.slot_ids = slot_ids; fx.partial = partial; fx.ldp = ldp; fx.base = base;
    fx.ldbase = ldbase; fx.S_out = S_out; fx.lds = lds; fx.M = M; fx.ldm = ldm; fx.transposed = transposed ? 1 : 0; fx.epilogue = epilogue;
    return emulate_any(2, nullptr, 0, nullptr, 0, nullptr, 0, n, k, N, W, ldw, C, ldc, 0, &fx);
}

extern "C" int pgcn_linear_epilogue_f32(const float *X, int64_t ldx, int64_t n, int32_t k, const float *W, int64_t ldw, int32_t wrows,
                                        int32_t wcols, int32_t transposed, const float *M, int64_t ldm, float *C, int64_t ldc,
                                        int32_t epilogue, void *) {
    if (wrows <= 0 || wcols <= 0 || (transposed ? wcols : wrows) != k || epilogue < 0 || epilogue > 2)
        return fail(-1, "pgcn_linear_epilogue_f32: W does not match the width of X / bad epilogue");
    const int N = transposed ? wrows : wcols;
    if (int rc = check(X, ldx, n, k, N, W, ldw, wrows, wcols, C, ldc)) return rc;
    if (epilogue == 2 && (!M || ldm < N)) return fail(-1, "pgcn_linear_epilogue_f32: the mask epilogue needs M");
    if (n == 0) return 0;
    // (the kernel's plain loader = load_tile; emulated through the fix-up path with every row taken from `base`: the same pieces)
    static RowFix *all_base = nullptr;
    static int64_t cap = 0;
    if (n > cap) { delete[] all_base; all_base = new RowFix[n]; cap = n; for (int64_t i = 0; i < n; ++i) all_base[i] = RowFix{0, -1}; }
    FixupArgs fx;
    fx.row_fix = all_base; fx.base = X; fx.ldbase = ldx; fx.M = M; fx.ldm = ldm; fx.transposed = transposed ? 1 : 0; fx.epilogue = epilogue;
    return emulate_any(2, nullptr, 0, nullptr, 0, nullptr, 0, n, k, N, W, ldw, C, ldc, 0, &fx);
}
