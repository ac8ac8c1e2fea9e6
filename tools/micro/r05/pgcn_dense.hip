// pgcn_dense.hip -- the dense products of a layer on the bf16 matrix cores at fp32 accuracy, fused with what surrounds them:
//     forward    Y  = relu(X . W^T)                  (/root/reference/GPU/PGCN.py:146-147  `F.relu(self.linear(AH))`)
//     backward   Gm = G (.) [Y > 0],  dX = Gm . W    (autograd of the same two lines; dW = Gm^T . X stays a library GEMM)
// north_star reserves MFMA for exactly this contraction; rounds 1-4 ran it as stock rocBLAS kernels (85 us per product at
// n = 232 965, f = 128: profiles/r04_gemm_pick.txt) plus a ReLU pass (35 us) and a mask pass (50 us).  An n x K x N product with
// K, N <= 128 moves 2 x n x 512 B and needs 2 n K N flops: HBM-bound (30 us at 8 TB/s for the benchmark layer) as long as the
// matrix pipes stay under that, which the fp32 MFMA does not (49 us at its peak) and six bf16 MFMAs do (18 us).
// Measured (MI355X, tools/micro/dense_fused_bench, profiles/r05_dense_fused_variants.txt): forward 63.3 us (3.8 TB/s), input
// gradient with Gm written 116.8 us (4.1 TB/s); f = 64: 22.9 / 41.1 us (5.2 / 5.8 TB/s).
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
// registers and runs nb x 6 MFMAs per k step against the image; the B operands of step t + 1 are read from LDS under the
// MFMAs of step t.  The forward keeps two register sets of tiles that swap roles: the next tile's loads go out once half of
// the current tile's MFMAs are done and land under the other half, its stores and the other wave of the SIMD.  The backward
// streams G and Y by half tiles, keeps G (.) [Y > 0] and writes it out as Gm for the weight gradient.  C leaves as 16 x nb
// non-temporal dword stores per lane, two full 128-byte lines per store.
//
// What was tried around this shape and lost (r04 / r05 harness runs, HISTORY.md section 9 and 10): a whole next tile in flight
// from the top of a tile (spills), no prefetch, loads after the stores, unpipelined LDS reads, the transposed accumulator tile
// with 16-byte stores (71.8 us), the last partial round of tiles spread over all CUs (64.1 us with the transposed tile).
//
// The index arithmetic lives in pgcn_dense_tile.h, which tests/native/pgcn_dense_emu.cpp also compiles for the host and runs lane
// by lane around an emulated MFMA (tests/test_zz_dense_fused.py); the MFMA operand layout itself is the one
// pgcn_spmm_dense3.hip runs on hardware.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#define PG_HD __device__ __forceinline__

namespace pgcn_dense {

#include "pgcn_dense_tile.h"

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// one partial product: x = a plane of the streamed operand (rows of X / G), w = a plane of the image
PG_HD f32x16 mma(const u32x4 &x, const u32x4 &w, const f32x16 &c) {
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
        for (int j = 0; j < 6; ++j) acc[nb] = mma(a[kPA[j]], b[t & 1][kPB[j]], acc[nb]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// k steps [KS0, KS1) of a tile's product
template <int NKS, int NBLK, int KS0, int KS1>
PG_HD void tile_product(const TileA<NKS> &t, const char *image, int lane, f32x16 (&acc)[NBLK]) {
    f32x4 v[KS1 - KS0][2];
#pragma unroll
    for (int i = 0; i < KS1 - KS0; ++i) { v[i][0] = t.v[KS0 + i][0]; v[i][1] = t.v[KS0 + i][1]; }
    product_steps<NBLK, KS1 - KS0>(v, image, lane, KS0, acc);
}
template <int NBLK>
PG_HD void zero_acc(f32x16 (&acc)[NBLK]) {
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
}

// C (n x N) = op(A) (n x K) . Bm (K x N); MASK: op(A) = A (.) [Y > 0] (and Gm = op(A) when Gm != nullptr); relu: C = relu(C);
// EMASK (not with MASK): C = the product where Y (n x N, here the mask of the OUTPUT) > 0, else 0.
template <int NKS, int NBLK, bool MASK, bool EMASK = false>
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
    int64_t tile = (int64_t)blockIdx.x * kWaves + w;          // first tile of this wave
    f32x16 acc[NBLK];
    if constexpr (MASK) {
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
            product_steps<NBLK, H>(v, image, lane, 0, acc);
            mask_half<H>(v, raw, Gm, ldgm, row0, n, K, lane, H);
            __builtin_amdgcn_sched_barrier(0);
            if (tn < ntiles) load_half<H>(raw, A, lda, Y, ldy, tn * kRows, n, K, lane, 0);    // under the second half's and the stores
            __builtin_amdgcn_sched_barrier(0);
            product_steps<NBLK, H>(v, image, lane, H, acc);
            store_c(acc, NBLK, C, ldc, row0, n, N, lane, relu);
            tile = tn;
        }
    } else {
        // two register sets that swap roles: `from` is multiplied while `into` receives the wave's next tile
        TileA<NKS> t0, t1;
        auto one_tile = [&](const TileA<NKS> &from, TileA<NKS> &into) {
            const int64_t tn = tile + stride;
            zero_acc(acc);
            tile_product<NKS, NBLK, 0, NKS / 2>(from, image, lane, acc);
            if (tn < ntiles) load_tile<NKS>(into, A, lda, tn * kRows, n, K, lane);
            __builtin_amdgcn_sched_barrier(0);
            tile_product<NKS, NBLK, NKS / 2, NKS>(from, image, lane, acc);
            if constexpr (EMASK) store_c_masked(acc, NBLK, C, ldc, Y, ldy, tile * kRows, n, N, lane);
            else store_c(acc, NBLK, C, ldc, tile * kRows, n, N, lane, relu);
            tile = tn;
        };
        if (tile < ntiles) load_tile<NKS>(t0, A, lda, tile * kRows, n, K, lane);
        while (tile < ntiles) {
            one_tile(t0, t1);
            if (tile >= ntiles) break;
            one_tile(t1, t0);
        }
    }
}

// ---- the same product with the fix-up of the aggregation as its loader (pgcn_dense_tile.h: sum_half) ------------------------------
PG_HD int wave_max(int x) {
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const int y = __shfl_xor(x, o);
        x = y > x ? y : x;
    }
    return x;
}

// C (n x N) = epi(S . Bm),  S[r] = the sum of row r's partial rows (or base[r]); S is written out when S_out != nullptr.
// EPI 0: none, 1: relu, 2: C = product where M > 0 else 0.  No tile is prefetched across the loop: a tile's loads depend on its
// rows' slot lists; the other seven waves of the workgroup cover a wave's round trips.
template <int NKS, int NBLK, int EPI>
__global__ __launch_bounds__(kThreads, 2) void fixup_dense_kernel(const RowFix *__restrict__ row_fix, const int32_t *__restrict__ slot_ids,
                                                                  const float *__restrict__ partial, int64_t ldp,
                                                                  const float *__restrict__ base, int64_t ldbase, int64_t n, int K, int N,
                                                                  const float *__restrict__ W, int64_t ldw, int transposed,
                                                                  float *__restrict__ S_out, int64_t lds, const float *__restrict__ M,
                                                                  int64_t ldm, float *__restrict__ C, int64_t ldc) {
    extern __shared__ __attribute__((aligned(16))) char image[];
    {
        constexpr int kMine = kSlotsPerPlane / kThreads;
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
    constexpr int H = NKS / 2;
    f32x16 acc[NBLK];
    for (int64_t tile = (int64_t)blockIdx.x * kWaves + w; tile < ntiles; tile += stride) {
        const int64_t row0 = tile * kRows, row = row0 + (lane & 31);
        RowFix rf = {0, 0};
        if (row < n) rf = row_fix[row];
        const int tmax = partial ? wave_max(rf.count) : 0;
        zero_acc(acc);
        f32x4 v[H][2];
        sum_half<H>(v, partial, ldp, slot_ids, base, ldbase, row < n ? row : 0, rf, tmax, K, lane, 0);
        if (S_out) store_half<H>(v, S_out, lds, row0, n, K, lane, 0);
        __builtin_amdgcn_sched_barrier(0);
        product_steps<NBLK, H>(v, image, lane, 0, acc);
        sum_half<H>(v, partial, ldp, slot_ids, base, ldbase, row < n ? row : 0, rf, tmax, K, lane, H);
        if (S_out) store_half<H>(v, S_out, lds, row0, n, K, lane, H);
        __builtin_amdgcn_sched_barrier(0);
        product_steps<NBLK, H>(v, image, lane, H, acc);
        if constexpr (EPI == 2) store_c_masked(acc, NBLK, C, ldc, M, ldm, row0, n, N, lane);
        else store_c(acc, NBLK, C, ldc, row0, n, N, lane, EPI);
    }
}

template <int NKS, int NBLK, int EPI>
int launch_fixup(const RowFix *row_fix, const int32_t *slot_ids, const float *partial, int64_t ldp, const float *base, int64_t ldbase,
                 int64_t n, int K, int N, const float *W, int64_t ldw, int transposed, float *S_out, int64_t lds, const float *M,
                 int64_t ldm, float *C, int64_t ldc, int workgroups, hipStream_t s) {
    auto kern = fixup_dense_kernel<NKS, NBLK, EPI>;
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
    hipLaunchKernelGGL(kern, dim3((unsigned)workgroups), dim3(kThreads), kImageBytes, s, row_fix, slot_ids, partial, ldp, base, ldbase, n, K,
                       N, W, ldw, transposed, S_out, lds, M, ldm, C, ldc);
    return hipGetLastError() == hipSuccess ? 0 : fail(-1, "kernel launch");
}

template <int EPI>
int dispatch_fixup(const RowFix *row_fix, const int32_t *slot_ids, const float *partial, int64_t ldp, const float *base, int64_t ldbase,
                   int64_t n, int K, int N, const float *W, int64_t ldw, int transposed, float *S_out, int64_t lds, const float *M,
                   int64_t ldm, float *C, int64_t ldc, hipStream_t s) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return fail(-1, "hipDeviceGetAttribute(MultiprocessorCount)");
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const int64_t need = (ntiles + kWaves - 1) / kWaves;
    const int wgs = (int)(need < cus ? need : cus);
    const int nks = (K + 15) / 16, nblk = (N + 31) / 32;
#define PGCN_DENSE_CASE(KS, NB)                                                                                              \
    if (nks <= KS && nblk <= NB)                                                                                             \
        return launch_fixup<KS, NB, EPI>(row_fix, slot_ids, partial, ldp, base, ldbase, n, K, N, W, ldw, transposed, S_out, lds, M, ldm, C, \
                                         ldc, wgs, s);
    PGCN_DENSE_CASE(4, 2)
    PGCN_DENSE_CASE(4, 4)
    PGCN_DENSE_CASE(8, 2)
    PGCN_DENSE_CASE(8, 4)
#undef PGCN_DENSE_CASE
    return fail(-2, "pgcn_fixup_linear_f32: widths above 128");
}

template <int NKS, int NBLK, bool MASK, bool EMASK = false>
int launch(const float *A, int64_t lda, const float *Y, int64_t ldy, float *Gm, int64_t ldgm, int64_t n, int K, int N,
           const float *W, int64_t ldw, int transposed, float *C, int64_t ldc, int relu, int workgroups, hipStream_t s) {
    auto kern = dense_kernel<NKS, NBLK, MASK, EMASK>;
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

template <bool MASK, bool EMASK = false>
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
        return launch<KS, NB, MASK, EMASK>(A, lda, Y, ldy, Gm, ldgm, n, K, N, W, ldw, transposed, C, ldc, relu, wgs, s);
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

// C (n x N, ldc) = epi(X . Bm):  X: n x k (ldx);  W: wrows x wcols (ldw);  transposed 1: Bm = W^T (N = wrows, wcols = k), 0: Bm = W
// (wrows = k, N = wcols);  epilogue 0: none, 1: relu, 2: keep where M (n x N, ldm) > 0, else 0 -- the input gradient of a layer
// with the ReLU mask of the layer below folded in.
extern "C" int pgcn_linear_epilogue_f32(const float *X, int64_t ldx, int64_t n, int32_t k, const float *W, int64_t ldw, int32_t wrows,
                                        int32_t wcols, int32_t transposed, const float *M, int64_t ldm, float *C, int64_t ldc,
                                        int32_t epilogue, void *stream) {
    using namespace pgcn_dense;
    if (wrows <= 0 || wcols <= 0 || (transposed ? wcols : wrows) != k || epilogue < 0 || epilogue > 2)
        return fail(-1, "pgcn_linear_epilogue_f32: W does not match the width of X / bad epilogue");
    const int N = transposed ? wrows : wcols;
    if (int rc = check(X, ldx, n, k, N, W, ldw, wrows, wcols, C, ldc)) return rc;
    if (epilogue == 2 && (!M || ldm < N)) return fail(-1, "pgcn_linear_epilogue_f32: the mask epilogue needs M");
    if (n == 0) return 0;
    if (epilogue == 2)
        return dispatch<false, true>(X, ldx, M, ldm, nullptr, 0, n, k, N, W, ldw, transposed ? 1 : 0, C, ldc, 0, (hipStream_t)stream);
    return dispatch<false>(X, ldx, nullptr, 0, nullptr, 0, n, k, N, W, ldw, transposed ? 1 : 0, C, ldc, epilogue, (hipStream_t)stream);
}

// C (n x N) = epi(S . Bm) with S[r] = the ordered sum of row r's partial rows -- csrc's fix-up folded into the dense product
// (pgcn_dense_tile.h: sum_half; bit-identical to pgcn_spmm_fixup_f32 followed by pgcn_linear_relu_f32).
//   row_fix: n x {begin, count} (count < 0: S[r] = base[r]);  slot_ids: the slot lists (NULL: slots begin .. begin + count);
//   partial: the producers' work-space, rows ldp floats apart;  k: width of S;  W: wrows x wcols (ldw);
//   transposed 1: Bm = W^T (W = nn.Linear's weight, N = wrows, wcols = k);  0: Bm = W (wrows = k, N = wcols);
//   S_out (n x k, lds): S written out when not NULL;  epilogue 0: none, 1: relu, 2: keep where M (n x N, ldm) > 0, else 0.
extern "C" int pgcn_fixup_linear_f32(const int32_t *row_fix, const int32_t *slot_ids, const float *partial, int64_t ldp,
                                     const float *base, int64_t ldbase, int64_t n, int32_t k, const float *W, int64_t ldw,
                                     int32_t wrows, int32_t wcols, int32_t transposed, float *S_out, int64_t lds, const float *M,
                                     int64_t ldm, float *C, int64_t ldc, int32_t epilogue, void *stream) {
    using namespace pgcn_dense;
    if (wrows <= 0 || wcols <= 0 || (transposed ? wcols : wrows) != k) return fail(-1, "pgcn_fixup_linear_f32: W does not match the width of S");
    const int N = transposed ? wrows : wcols;
    if (int rc = check_fixup(row_fix, partial, ldp, base, ldbase, n, k, N, W, ldw, wcols, S_out, lds, M, ldm, C, ldc, epilogue)) return rc;
    if (n == 0) return 0;
    const RowFix *rf = reinterpret_cast<const RowFix *>(row_fix);
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case 0: return dispatch_fixup<0>(rf, slot_ids, partial, ldp, base, ldbase, n, k, N, W, ldw, transposed ? 1 : 0, S_out, lds, M, ldm, C, ldc, s);
        case 1: return dispatch_fixup<1>(rf, slot_ids, partial, ldp, base, ldbase, n, k, N, W, ldw, transposed ? 1 : 0, S_out, lds, M, ldm, C, ldc, s);
        default: return dispatch_fixup<2>(rf, slot_ids, partial, ldp, base, ldbase, n, k, N, W, ldw, transposed ? 1 : 0, S_out, lds, M, ldm, C, ldc, s);
    }
}
