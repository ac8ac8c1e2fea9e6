// pgcn_dense.hip -- the dense products of a layer on the bf16 matrix cores at fp32 accuracy, fused with what surrounds them:
//     forward    Y  = relu(X . W^T)  (+ the sign mask of Y)     (/root/reference/GPU/PGCN.py:146-147  `F.relu(self.linear(AH))`)
//     backward   Gm = G (.) [Y > 0],  dX = Gm . W               (autograd of the same two lines; dW = Gm^T . X: pgcn_wgrad.hip)
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
// conflicts).  Bm = W^T (forward: Bm[k][o] = W[o][k]) or W (backward: Bm[o][k] = W[o][k]).  A wave owns 32-row tiles of the
// streamed operand (tile t = 8 workgroup + wave, stride 8 workgroups): lane (lo, hi) reads row lo, columns 16 ks + 8 hi .. + 8 of
// k step ks as two 16-byte loads (a PIECE), straight into A-operand registers.
//
// Schedule (r06).  The r05 kernel ran a tile as  loads | 8 x (split, 4 chains of six MFMAs) | 64 dword stores: measured 63 / 117 us
// where the matrix pipe needs ~29 us at the clock it sustains (1.5 GHz) and the memory ~40 -- the phases of the eight waves of
// a workgroup line up (they start together and do the same work), so the store tail (issue-bound: 64 stores per lane), the loads
// and the MFMAs take turns instead of overlapping (profiles/r05_dense_fused_variants.txt: 10 us return with the stores removed).
// Now a tile is multiplied COLUMN BLOCK by column block (outer nb, inner ks: 8 chains on one accumulator, the next block on the
// other), from the bf16 planes of ALL its pieces held in registers (96), and everything else is spread between the chains:
//   * the 16 stores of block nb - 1 (and, forward, its sign-mask word) go out under the chains of block nb, two per chain;
//   * during the LAST block of a tile the pieces of the wave's NEXT tile (in flight since one tile ago, 64 registers) are masked
//     (backward), written out as Gm, split into the plane registers piece ks - 1 has just stopped using, and the piece of the
//     tile after that is requested into the freed registers: every load has a whole tile (~4 us) to land;
//   * the B operands of chain t + 1 are read from LDS before chain t.
// So a wave's instruction stream is one uniform sequence of [chain of six MFMAs, ~10-70 other instructions], with no load or
// store phase for the waves of a CU to line up in.
// The backward no longer reads Y: the forward leaves a sign mask (1 bit per element: 3.7 MB instead of 119 MB at the benchmark
// layer; pgcn_dense_tile.h) -- which is also what lets a whole tile of G be in flight in 64 registers instead of 128.
//
// What was tried around the r05 shape and lost (harness runs, HISTORY.md section 9 and 10): a whole next tile in flight from the
// top of a tile (spills), no prefetch, loads after the stores, unpipelined LDS reads, the transposed accumulator tile with
// 16-byte stores (71.8 us), the last partial round of tiles spread over all CUs (64.1 us with the transposed tile).
//
// The index arithmetic lives in pgcn_dense_tile.h, which tests/native/pgcn_dense_emu.cpp also compiles for the host and runs lane
// by lane around an emulated MFMA (tests/test_zz_dense_fused.py); the MFMA operand layout itself is the one
// pgcn_spmm_dense3.hip runs on hardware.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#define PG_HD __device__ __forceinline__

namespace pgcn_dense {

#include "pgcn_dense_tile.h"

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int I, int N, class F>
PG_HD void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// one partial product: x = a plane of the streamed operand (rows of X / G), w = a plane of the image
PG_HD f32x16 mma(const u32x4 &x, const u32x4 &w, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), c, 0, 0, 0);
}
PG_HD void read_b(u32x4 (&b)[3], const char *image, int ks, int nb, int lane) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4 *>(image + image_offset(pl, ks, nb, lane));
}

// The sign-mask bits of accumulator register R over the wave: lane R of `word` takes the ballot's low half (the rows of half wave 0),
// lane 16 + R its high half.  One asm block, because the compare -> v_writelane hand-over through VCC needs wait states that the
// compiler's hazard recogniser does not insert around inline asm (first hardware run of r06: v_writelane issued straight after the
// v_cmp read the PREVIOUS compare's VCC -- 20 % of the mask bits belonged to the neighbouring register).
template <int R>
PG_HD void mask_lanes(uint32_t &word, float x) {
    asm volatile("v_cmp_nge_f32_e32 vcc, 0, %1\n\ts_nop 4\n\tv_writelane_b32 %0, vcc_lo, %2\n\tv_writelane_b32 %0, vcc_hi, %3"
                 : "+v"(word)
                 : "v"(x), "n"(R), "n"(16 + R)
                 : "vcc");
}

struct Args {
    const float *A;            // the streamed operand: X (forward) or G (input gradient), n x K
    int64_t lda;
    const uint32_t *mask_in;   // input gradient: the sign mask of the forward's output (n x mask_words(K)), or NULL: no mask
    float *Gm;                 // input gradient: G (.) mask written out when not NULL (may be A itself)
    int64_t ldgm;
    uint32_t *mask_out;        // forward: the sign mask of C (n x mask_words(N)) when not NULL
    int64_t n;
    int K, N;
    const float *W;
    int64_t ldw;
    int transposed, wvec;
    float *C;
    int64_t ldc;
    int relu;
};

// C (n x N) = op(A) (n x K) . Bm (K x N).  MODE 0: op(A) = A, C = relu ? max(C, 0) : C, mask_out = sign mask of the product;
// MODE 1: op(A) = A where mask_in says so (all of A when mask_in == NULL), written out as Gm when asked for.
template <int NKS, int NBLK, int MODE, bool RAGGED>
__global__ __launch_bounds__(kThreads, 2) void dense_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char image[];
    {
        constexpr int kMine = kSlotsPerPlane / kThreads;      // slots s = thread + 512 q: k step (thread >> 8) + 2 q, column block per wave
        float v[kMine][8];
        const bool vec = a.wvec != 0;
#pragma unroll
        for (int q = 0; q < kMine; ++q) {
            const int s = (int)threadIdx.x + q * kThreads;
            if ((s >> 8) < NKS && ((s >> 6) & 3) < NBLK) slot_load(a.W, a.ldw, a.transposed, a.K, a.N, s, v[q], vec);
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
    const int64_t n = a.n, ntiles = (n + kRows - 1) / kRows;
    const int64_t stride = (int64_t)gridDim.x * kWaves;
    int64_t tile = (int64_t)blockIdx.x * kWaves + w;          // first tile of this wave
    if (tile >= ntiles) return;
    const int K = a.K, N = a.N, mwK = mask_words(K), mwN = mask_words(N);
    constexpr int T = NKS * NBLK, kChunk = 16 / NKS;          // steps of a tile; accumulator registers stored per step

    u32x4 ap[NKS][3];                                         // bf16 planes of the current tile
    f32x4 raw[NKS][2];                                        // pieces of the wave's next tile (in flight)
    uint32_t mwn[4] = {~0u, ~0u, ~0u, ~0u};                   // its row's mask words (MODE 1)
    f32x16 acc[2];

    // Lane offsets inside a tile's window (fixed for the whole kernel) and the scalar row offsets of the accumulator registers
    const uint32_t a_off = piece_lane_offset(a.lda, lane), gm_off = piece_lane_offset(a.ldgm, lane);
    const uint32_t c_off = acc_lane_offset(a.ldc, lane), c_row = (uint32_t)(a.ldc * 4);
    const uint32_t mi_off = (uint32_t)((lane & 31) * mwK * 4);                       // the lane's row of mask_in
    const uint32_t mo_off = (uint32_t)(mask_row_of_lane(lane & 31) * mwN * 4);        // the row whose mask_out word lane < 32 collects
    // windows of tile t (t wave-uniform): the rows of each matrix from the tile's first row on.  A tile beyond the last one, or a
    // matrix the caller did not pass, has an EMPTY window: its loads return zero and its stores are dropped without touching
    // memory -- which is what keeps the loop body free of branches (a branch around a load makes the compiler wait for every
    // outstanding store at the next use: vmcnt(0))
    auto win_of = [&](const void *M, int64_t ld, int64_t t, int width) {
        return tile_window(M, ld, M ? n : 0, t * kRows, width);
    };
    auto win_a = [&](int64_t t) { return win_of(a.A, a.lda, t, K); };
    auto win_gm = [&](int64_t t) { return win_of(a.Gm, a.ldgm, t, K); };
    auto win_c = [&](int64_t t) { return win_of(a.C, a.ldc, t, N); };
    auto win_mi = [&](int64_t t) { return win_of(a.mask_in, mwK, t, mwK); };
    auto win_mo = [&](int64_t t) { return win_of(a.mask_out, mwN, t, mwN); };
    auto load_tile = [&](window_t wa, int ks) { load_piece<RAGGED>(raw[ks], wa, a_off, ks, K, lane); };
    const uint32_t no_mask = a.mask_in ? 0u : ~0u;            // (no mask passed: every bit set)
    auto load_mask_words = [&](window_t wm) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mwn[q] = (q < mwK ? win_load4u(wm, mi_off + 4 * (q < mwK ? q : 0), 0) : 0u) | no_mask;
        }
    };
    // piece ks of tile t has landed in raw[ks]: mask it, write it out as Gm, split it into ap[ks]
    auto consume = [&](window_t wgm, int ks) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) raw[ks][h] = mask4_bits(raw[ks][h], mwn[ks >> 1], ks, h, lane);
            store_piece<RAGGED>(raw[ks], wgm, gm_off, ks, K, lane);
        }
        split8(raw[ks][0], raw[ks][1], ap[ks]);
    };

    // prologue: the first tile through the same path, nothing to overlap it with yet
    {
        const window_t wa = win_a(tile), wgm = win_gm(tile);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) load_tile(wa, ks);
        load_mask_words(win_mi(tile));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) consume(wgm, ks);
    }
    int64_t tn = tile + stride;
    {
        const window_t wa = win_a(tn);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) load_tile(wa, ks);
        load_mask_words(win_mi(tn));
    }
    window_t wc_prev = win_c(ntiles), wmo_prev = win_mo(ntiles);  // of the tile whose last block is still to be stored: none yet (empty)
    uint32_t mword = 0;                                       // the mask word this lane collects for the block being stored

    // registers [kChunk c, kChunk c + kChunk) of column block nb of tile t: their stores and their share of the sign mask
    auto store_chunk = [&](const f32x16 &x, auto cc, int nb, window_t wc, window_t wmo) {
        constexpr int c = decltype(cc)::value;
        store_regs<RAGGED>(x, kChunk * c, kChunk, nb, wc, c_off, c_row, N, lane, a.relu);
        if constexpr (MODE == 0) {
            static_for<kChunk * c, kChunk * (c + 1)>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                mask_lanes<r>(mword, x[r]);               // (mask_bit_of: !(x <= 0))
            });
            // (bits of columns beyond N come out of zero products: 0 > 0 is false; lanes 32..63 and blocks beyond the mask's
            //  width aim outside the window)
            if constexpr (c == NKS - 1) win_store4u(wmo, (lane < 32 && nb < mwN) ? mo_off + 4 * nb : 0xfffffff0u, 0, mword);
        }
    };

    u32x4 b[2][3];
    read_b(b[0], image, 0, 0, lane);
    while (true) {
        const int64_t tn2 = tn + stride;
        const bool has_next = tn < ntiles;
        // the windows of this trip, once (scalar work): C and the mask of the current tile, Gm of the next, A and the mask of the one after
        const window_t wc = win_c(tile), wmo = win_mo(tile), wgm = win_gm(tn), wa2 = win_a(tn2), wmi2 = win_mi(tn2);
        static_for<0, T>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int nb = t / NKS, ks = t % NKS, P = nb & 1;
            constexpr int t1 = (t + 1) % T;                   // (the image does not change: step 0 of the next tile is read at the end of this one)
            read_b(b[(t + 1) & 1], image, t1 % NKS, t1 / NKS, lane);
            __builtin_amdgcn_sched_barrier(0);
            {
                PGCN_DENSE_PRODUCTS;
                if constexpr (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[P][r] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[P] = mma(ap[ks][kPA[j]], b[t & 1][kPB[j]], acc[P]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // between the chains: the stores of the block that was finished one pass ago ...
            if constexpr (nb > 0) store_chunk(acc[P ^ 1], std::integral_constant<int, ks>{}, nb - 1, wc, wmo);
            else store_chunk(acc[1], std::integral_constant<int, ks>{}, NBLK - 1, wc_prev, wmo_prev);
            // ... and, in the last pass, the next tile's piece ks - 1 (its plane registers are free now) and the load behind it
            if constexpr (nb == NBLK - 1 && ks >= 1) {
                consume(wgm, ks - 1);
                load_tile(wa2, ks - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        consume(wgm, NKS - 1);
        load_tile(wa2, NKS - 1);
        load_mask_words(wmi2);
        wc_prev = wc;
        wmo_prev = wmo;
        if (!has_next) break;
        tile = tn;
        tn = tn2;
    }
    // the last block of the wave's last tile
    static_for<0, NKS>([&](auto cc) { store_chunk(acc[1], cc, NBLK - 1, wc_prev, wmo_prev); });
}

// ---- the sign mask of an existing matrix (callers that hold Y but not the forward's mask: tests, the library-GEMM route) ---------------
__global__ __launch_bounds__(256) void sign_mask_kernel(const float *__restrict__ Y, int64_t ldy, int64_t n, int N, uint32_t *__restrict__ mask) {
    const int mw = (N + 31) / 32;
    const int64_t word = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread per mask word
    if (word >= n * mw) return;
    const int64_t row = word / mw;
    const int w = (int)(word % mw);
    uint32_t bits = 0;
    for (int bidx = 0; bidx < 32; ++bidx) {
        const int c = 32 * w + bidx;
        if (c < N && mask_bit_of(Y[row * ldy + c])) bits |= 1u << bidx;
    }
    mask[word] = bits;
}

template <int NKS, int NBLK, int MODE, bool RAGGED>
int launch(const Args &a, int workgroups, hipStream_t s) {
    auto kern = dense_kernel<NKS, NBLK, MODE, RAGGED>;
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
    hipLaunchKernelGGL(kern, dim3((unsigned)workgroups), dim3(kThreads), kImageBytes, s, a);
    return hipGetLastError() == hipSuccess ? 0 : fail(-1, "kernel launch");
}

template <int MODE>
int dispatch(Args a, hipStream_t s) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return fail(-1, "hipDeviceGetAttribute(MultiprocessorCount)");
    const int64_t ntiles = (a.n + kRows - 1) / kRows;
    const int64_t need = (ntiles + kWaves - 1) / kWaves;
    const int wgs = (int)(need < cus ? need : cus);           // one persistent workgroup per CU (96 KB of LDS each)
    const int nks = (a.K + 15) / 16, nblk = (a.N + 31) / 32;
    a.wvec = slot_vec_ok(a.W, a.ldw, a.transposed, a.K) ? 1 : 0;
#define PGCN_DENSE_CASE(KS, NB)                                                                                              \
    if (nks <= KS && nblk <= NB) {                                                                                           \
        if (a.K == 16 * KS && a.N == 32 * NB) return launch<KS, NB, MODE, false>(a, wgs, s);                                  \
        return launch<KS, NB, MODE, true>(a, wgs, s);                                                                        \
    }
    PGCN_DENSE_CASE(4, 2)
    PGCN_DENSE_CASE(4, 4)
    PGCN_DENSE_CASE(8, 2)
    PGCN_DENSE_CASE(8, 4)
#undef PGCN_DENSE_CASE
    return fail(-2, "pgcn_dense: widths above 128");
}

}  // namespace pgcn_dense

// ---- the C ABI (include/pgcn_gemm.h) ---------------------------------------------------------------------------------------
extern "C" const char *pgcn_dense_last_error(void) { return pgcn_dense::g_err; }

// Y (n x fout, ldy) = [relu] (X (n x fin, ldx) . W^T),  W: fout x fin row-major (nn.Linear's weight), on `stream`;
// mask (optional, n x ceil(fout / 32) words): bit b of word [row][w] = (Y[row][32 w + b] > 0).
// 0; -2: shape / alignment outside what the kernel takes (the caller uses the library GEMM); -1: errors.
extern "C" int pgcn_linear_relu_f32(const float *X, int64_t ldx, int64_t n, int32_t fin, const float *W, int64_t ldw, int32_t fout,
                                    float *Y, int64_t ldy, int32_t relu, uint32_t *mask, void *stream) {
    using namespace pgcn_dense;
    if (int rc = check(X, ldx, n, fin, fout, W, ldw, fout, fin, Y, ldy)) return rc;
    if (n == 0) return 0;
    Args a{};
    a.A = X; a.lda = ldx; a.mask_out = mask; a.n = n; a.K = fin; a.N = fout; a.W = W; a.ldw = ldw; a.transposed = 1; a.C = Y; a.ldc = ldy;
    a.relu = relu ? 1 : 0;
    return dispatch<0>(a, (hipStream_t)stream);
}

// Gm = G where the mask says Y > 0, else 0 (written when Gm != NULL; may be G itself),  dX (n x fin, lddx) = Gm . W;  G, Gm: n x fout;
// mask: the forward's sign mask (n x ceil(fout / 32) words; NULL: Gm = G).
extern "C" int pgcn_linear_relu_grad_input_f32(const float *G, int64_t ldg, const uint32_t *mask, float *Gm, int64_t ldgm, int64_t n,
                                               int32_t fout, const float *W, int64_t ldw, int32_t fin, float *dX, int64_t lddx,
                                               void *stream) {
    using namespace pgcn_dense;
    if (int rc = check(G, ldg, n, fout, fin, W, ldw, fout, fin, dX, lddx)) return rc;
    if (Gm && (ldgm % 4 || (uintptr_t)Gm % 16 || ldgm < fout)) return fail(-2, "pgcn_dense: rows of Gm must be 16-byte pieces");
    if (n == 0) return 0;
    Args a{};
    a.A = G; a.lda = ldg; a.mask_in = mask; a.Gm = Gm; a.ldgm = ldgm; a.n = n; a.K = fout; a.N = fin; a.W = W; a.ldw = ldw; a.transposed = 0;
    a.C = dX; a.ldc = lddx; a.relu = 0;
    return dispatch<1>(a, (hipStream_t)stream);
}

// mask (n x ceil(N / 32) words) = the sign mask of Y (n x N, ldy) in the layout of the two entry points above -- for callers that hold
// a forward output made elsewhere (the library-GEMM route, tests).
extern "C" int pgcn_sign_mask_f32(const float *Y, int64_t ldy, int64_t n, int32_t N, uint32_t *mask, void *stream) {
    using namespace pgcn_dense;
    if (n < 0 || N <= 0 || ldy < N || (n > 0 && (!Y || !mask))) return fail(-1, "pgcn_sign_mask_f32: bad argument");
    if (n == 0) return 0;
    const int64_t words = n * ((N + 31) / 32);
    hipLaunchKernelGGL(sign_mask_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Y, ldy, n, N, mask);
    return hipGetLastError() == hipSuccess ? 0 : fail(-1, "kernel launch (sign_mask_kernel)");
}
