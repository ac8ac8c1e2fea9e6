// pgcn_wgrad.hip -- the weight gradient of a layer, dW = Gm^T . X (fout x fin, contracted over the n rows), on the bf16 matrix
// cores at fp32 accuracy: the third dense product of `F.relu(self.linear(AH))` (/root/reference/GPU/PGCN.py:146-147; autograd's
// `grad_output.t() @ input`).  Rounds 1-5 ran it as a 64-slab batched library GEMM + a 64-way sum (72 + 22 us under the profiler at
// n = 232 965, f = 128, against 30 us to read both operands once at 8 TB/s); r04's first own kernel (tools/experiments/pgcn_wgrad.hip)
// ran 565 us: one 16-row step of prefetch against an HBM latency of microseconds, 64-bit address arithmetic and selects on every
// element.  This is the r06 rewrite.
//
// Shape of the problem: M = fout, N = fin <= 128 (at most 4 x 4 blocks of 32 x 32), K = n ~ 10^5..10^6.  For
// v_mfma_f32_32x32x16_bf16 the contraction index (rows r of Gm and X) must run along a lane's registers: lane (lo, hi) holds
//     A[i = o][kk = 8 hi + j] = Gm[r0 + 8 hi + j][32 ob + lo],   B[kk = 8 hi + j][jj = k] = X[r0 + 8 hi + j][32 kb + lo],  j = 0..7
// -- one column, eight consecutive rows: dword loads that are coalesced ACROSS lanes (32 consecutive columns = one 128-byte line
// per row and half wave), every byte of both matrices used, no transpose anywhere.
//
// Work split.  A wave owns a 64 x 64 tile of dW = 2 x 2 blocks (64 accumulator registers) and a contiguous range of 16-row steps;
// per step it loads 2 + 2 operand blocks (32 dwords per lane), splits them into the three bf16 planes (pgcn_dense_common.h) and
// issues 4 x 6 MFMAs.  Operands are requested kDepth steps ahead (kDepth x 32 registers in flight per lane: ~3 us of cover at
// the matrix rate), addressed as ONE scalar base per matrix and step plus eight lane offsets that never change -- no vector
// address arithmetic in the loop.  A 512-thread workgroup = two halves of four waves; a half covers the ceil(NOB / 2) x
// ceil(NKB / 2) tiles of dW (waves left over take further row ranges), the two halves take different row ranges and are added
// through LDS at the end, so a workgroup writes one partial matrix per row range of a half.  wgrad_sum_kernel adds the partial
// matrices in a fixed two-level order (16 groups of consecutive partials, then the 16 group sums): deterministic, no atomics.
//
// Arithmetic: x = x1 + x2 + x3 exactly (bf16 planes), the six partial products that matter smallest first, fp32 accumulation
// inside the MFMA -- the error class of an fp32 dot product (tests hold it to 1e-6 of sum |g||x|).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#define PG_HD __device__ __forceinline__

namespace pgcn_wgrad {

#include "pgcn_dense_common.h"

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kThreads = 512;
constexpr int kStepRows = 16;            // rows of one MFMA step (its k dimension)
constexpr int kMaxF = 128;
constexpr int kDepthBf16 = 3;            // steps of operands in flight (register sets of 32)
constexpr int kDepthF32 = 3;
constexpr int kGroups = 16;              // wgrad_sum_kernel: groups of consecutive partial matrices
#ifndef PGCN_WGRAD_F32MFMA
#define PGCN_WGRAD_F32MFMA 0             // 1: the library's entry point runs the fp32-MFMA form of the step
#endif

inline thread_local char g_err[256] = "";
inline int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

struct StepRaw {                          // the 32 values of one lane and step: operand blocks (a0, a1 of Gm; b0, b1 of X) x 8 rows
    float a[2][8], b[2][8];
};

// ---- loads: one buffer descriptor per matrix and wave (its row range; rows beyond the matrix read as zero by the hardware's bounds
// check), a scalar byte offset advanced per step, a lane's eight byte offsets (fixed), + 128 bytes for the second block of a tile.
// WGUARD: ragged widths -- a lane whose column lies beyond the width reads a clamped column and drops the value.
using rsrc_t = __amdgpu_buffer_rsrc_t;
PG_HD float buf_load(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
template <bool A1, bool B1, bool WGUARD>
PG_HD void load_step(StepRaw &s, rsrc_t g, rsrc_t x, uint32_t soff_g, uint32_t soff_x, const uint32_t (&goff)[8], const uint32_t (&xoff)[8],
                     bool a0_ok, bool a1_ok, bool b0_ok, bool b1_ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s.a[0][j] = buf_load(g, goff[j], soff_g);
        s.a[1][j] = A1 ? buf_load(g, goff[j] + ((!WGUARD || a1_ok) ? 128u : 0u), soff_g) : 0.f;
        s.b[0][j] = buf_load(x, xoff[j], soff_x);
        s.b[1][j] = B1 ? buf_load(x, xoff[j] + ((!WGUARD || b1_ok) ? 128u : 0u), soff_x) : 0.f;
        if constexpr (WGUARD) {
            s.a[0][j] = a0_ok ? s.a[0][j] : 0.f;
            s.a[1][j] = a1_ok ? s.a[1][j] : 0.f;
            s.b[0][j] = b0_ok ? s.b[0][j] : 0.f;
            s.b[1][j] = b1_ok ? s.b[1][j] : 0.f;
        }
    }
}

PG_HD void planes_of(const float (&v)[8], u32x4 (&p)[3]) {
    const f32x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
    split8(lo4, hi4, p);
}
PG_HD f32x16 mma(const u32x4 &x, const u32x4 &w, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), c, 0, 0, 0);
}

// acc[ai][bi] += A_ai^T-block . B_bi-block of one step.  The planes of the operand a chain does not need yet are split between the chains
// (VALU beside the other wave's MFMAs of the SIMD), never inside a chain of six.
template <bool A1, bool B1>
PG_HD void step_product(const StepRaw &s, f32x16 (&acc)[2][2]) {
    PGCN_DENSE_PRODUCTS;
    u32x4 a0[3], a1[3], b0[3], b1[3];
    planes_of(s.a[0], a0);
    planes_of(s.b[0], b0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[0][0] = mma(a0[kPA[i]], b0[kPB[i]], acc[0][0]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (B1) {
        planes_of(s.b[1], b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[0][1] = mma(a0[kPA[i]], b1[kPB[i]], acc[0][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (A1) {
        planes_of(s.a[1], a1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[1][0] = mma(a1[kPA[i]], b0[kPB[i]], acc[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (B1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[1][1] = mma(a1[kPA[i]], b1[kPB[i]], acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The same step on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 cycles, k = 2): lane (lo, hi) holds A[lo][hi], so
// the pair of rows (j, 8 + j) of the step that registers j of the two half waves hold IS one MFMA's operand -- no split, no VALU.
// 2.7 x the matrix-pipe cycles of the six bf16 products; which one wins on hardware is measured (tools/micro/dense_fused_bench).
template <bool A1, bool B1>
PG_HD void step_product_f32(const StepRaw &s, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.a[0][j], s.b[0][j], acc[0][0], 0, 0, 0);
        if constexpr (B1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.a[0][j], s.b[1][j], acc[0][1], 0, 0, 0);
        if constexpr (A1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.a[1][j], s.b[0][j], acc[1][0], 0, 0, 0);
        if constexpr (A1 && B1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.a[1][j], s.b[1][j], acc[1][1], 0, 0, 0);
    }
}
template <bool A1, bool B1, bool F32>
PG_HD void step_any(const StepRaw &s, f32x16 (&acc)[2][2]) {
    if constexpr (F32) step_product_f32<A1, B1>(s, acc);
    else step_product<A1, B1>(s, acc);
}

// steps [s0, s1) of the rows, contiguous per row range: range g of R takes steps [g S / R, (g + 1) S / R)
PG_HD void steps_of_range(int64_t nsteps, int ranges, int g, int64_t &s0, int64_t &s1) {
    s0 = nsteps * g / ranges;
    s1 = nsteps * (g + 1) / ranges;
}

// partial[range][FO x FI] (FO = 64 TO, FI = 64 TK) += this wave's tile.  TO x TK tiles of 64 x 64 per half; a half of four waves
// holds 4 / (TO TK) row ranges; halves h = 0, 1 of a workgroup take different row ranges and are added through LDS.
template <int TO, int TK, bool A1, bool B1, bool F32, bool WGUARD>
__global__ __launch_bounds__(kThreads, 2) void wgrad_kernel(const float *__restrict__ Gm, int64_t ldg, const float *__restrict__ X,
                                                            int64_t ldx, int64_t n, int fout, int fin, float *__restrict__ partial) {
    constexpr int kDepth = F32 ? kDepthF32 : kDepthBf16;         // register sets of operands in flight
    constexpr int kTiles = TO * TK, kSub = 4 / kTiles;            // tiles of dW, row ranges of a half
    constexpr int FI = 64 * TK, FO = 64 * TO;
    __shared__ float other[4 * 64 * 64];                          // the accumulators of half 1: [wave of the half][register][lane]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = w >> 2, wq = w & 3;
    const int tile = wq % kTiles, sub = wq / kTiles;
    const int to = tile / TK, tk = tile % TK;
    const int lo = lane & 31, hi = lane >> 5;
    // row ranges: (workgroup, half, sub)
    const int ranges = (int)gridDim.x * 2 * kSub;
    const int range = ((int)blockIdx.x * 2 + half) * kSub + sub;
    const int64_t nsteps = (n + kStepRows - 1) / kStepRows;
    int64_t s0, s1;
    steps_of_range(nsteps, ranges, range, s0, s1);
    const int ocol = 64 * to + lo, kcol = 64 * tk + lo;           // first-block columns of this lane
    const bool a0_ok = ocol < fout, a1_ok = ocol + 32 < fout, b0_ok = kcol < fin, b1_ok = kcol + 32 < fin;
    // lane offsets (bytes): row 8 hi + j of the step, the lane's column (clamped into the matrix for the guarded loads)
    uint32_t goff[8], xoff[8];
    {
        const int oc = a0_ok ? ocol : fout - 1, kc = b0_ok ? kcol : fin - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            goff[j] = (uint32_t)(((int64_t)(8 * hi + j) * ldg + oc) * 4);
            xoff[j] = (uint32_t)(((int64_t)(8 * hi + j) * ldx + kc) * 4);
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this wave's row range as two buffers: base = first row of the range, size = what is left of the matrix (clipped to the 32-bit
    // range of a descriptor: the host checked that a range's own bytes fit)
    const int64_t gstep = (int64_t)kStepRows * ldg * 4, xstep = (int64_t)kStepRows * ldx * 4;
    auto descriptor = [&](const float *M, int64_t ld, int64_t step_bytes) -> rsrc_t {
        const int64_t first = s0 * kStepRows;
        int64_t bytes = (n - first) * ld * 4;
        if (bytes > 0xfffff000LL) bytes = 0xfffff000LL;
        if (bytes < 0) bytes = 0;
        const uint64_t base = (uint64_t)(reinterpret_cast<const char *>(M) + s0 * step_bytes);
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)bhi << 32) | blo), 0, (int)nb, 0x00020000);
    };
    const rsrc_t gr = descriptor(Gm, ldg, gstep), xr = descriptor(X, ldx, xstep);
    const uint32_t gs = (uint32_t)gstep, xs = (uint32_t)xstep;
    const int nst = __builtin_amdgcn_readfirstlane((int)(s1 - s0));          // steps of this wave (the last one may be ragged: zeros)
    {
        // kDepth register sets, the set consumed in iteration i is refilled with step i + kDepth
        StepRaw raw[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d)
            if (d < nst) load_step<A1, B1, WGUARD>(raw[d], gr, xr, d * gs, d * xs, goff, xoff, a0_ok, a1_ok, b0_ok, b1_ok);
        int t = 0;
        uint32_t og = kDepth * gs, ox = kDepth * xs;                 // byte offsets of step t + kDepth
        while (t + kDepth <= nst) {                                  // kDepth steps per trip, every set refilled right after its use
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                step_any<A1, B1, F32>(raw[d], acc);
                if (t + d + kDepth < nst) load_step<A1, B1, WGUARD>(raw[d], gr, xr, og + d * gs, ox + d * xs, goff, xoff, a0_ok, a1_ok, b0_ok, b1_ok);
            }
            t += kDepth;
            og += kDepth * gs;
            ox += kDepth * xs;
        }
#pragma unroll
        for (int d = 0; d < kDepth; ++d)                             // the last nst - t < kDepth steps (already loaded)
            if (t + d < nst) step_any<A1, B1, F32>(raw[d], acc);
    }
    // the two halves of the workgroup: half 1 hands its accumulators over through LDS
    if (half == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) other[(wq * 64 + (i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (half == 0) {
        // partial matrix of (workgroup, sub): element (o, k) at [o * FI + k]; accumulator register r of lane (lo, hi), block (i, j) of
        // the tile = element (64 to + 32 i + (r & 3) + 8 (r >> 2) + 4 hi, 64 tk + 32 j + lo)
        float *out = partial + ((int64_t)blockIdx.x * kSub + sub) * (FO * FI);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 1 && !A1) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1 && !B1) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = 64 * to + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi, k = 64 * tk + 32 * j + lo;
                    out[o * FI + k] = acc[i][j][r] + other[(wq * 64 + (i * 2 + j) * 16 + r) * 64 + lane];
                }
            }
        }
    }
}

// dW[o][k] = sum over the partial matrices p = 0 .. parts - 1 of partial[p][o * FI + k], as kGroups sums of consecutive partials
// (each in order) added in group order.  Thread (e, g): four consecutive k of one o, group g; block = 16 chunks x 16 groups.
__global__ __launch_bounds__(256) void wgrad_sum_kernel(const float *__restrict__ partial, int parts, int FO, int FI, int fout, int fin,
                                                        float *__restrict__ dW, int64_t lddw) {
    __shared__ f32x4 part[16][kGroups + 1];
    const int g = threadIdx.x & (kGroups - 1), c = threadIdx.x >> 4;
    const int chunk = blockIdx.x * 16 + c;                         // float4 chunk of the FO x FI matrix
    const int per = (parts + kGroups - 1) / kGroups;
    const int p0 = g * per, p1 = (p0 + per < parts) ? p0 + per : parts;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (chunk * 4 < FO * FI) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(partial) + chunk;
        const int64_t slab = (int64_t)FO * FI / 4;
        int p = p0;
        for (; p + 4 <= p1; p += 4) {                              // four loads in flight, added in order
            const f32x4 x0 = src[(int64_t)p * slab], x1 = src[(int64_t)(p + 1) * slab], x2 = src[(int64_t)(p + 2) * slab],
                        x3 = src[(int64_t)(p + 3) * slab];
            s = ((s + x0) + x1) + x2;
            s = s + x3;
        }
        for (; p < p1; ++p) s = s + src[(int64_t)p * slab];
    }
    part[c][g] = s;
    __syncthreads();
    if (g == 0 && chunk * 4 < FO * FI) {
        f32x4 t = part[c][0];
#pragma unroll
        for (int q = 1; q < kGroups; ++q) t = t + part[c][q];
        const int o = (chunk * 4) / FI, k = (chunk * 4) % FI;
        if (o < fout) {
            float *dst = dW + (int64_t)o * lddw + k;
            if (k + 0 < fin) dst[0] = t.x;
            if (k + 1 < fin) dst[1] = t.y;
            if (k + 2 < fin) dst[2] = t.z;
            if (k + 3 < fin) dst[3] = t.w;
        }
    }
}

struct Shape {
    int TO, TK;          // 64 x 64 tiles of dW
    bool A1, B1;         // second block of a tile present (widths above 32 / 96)
    int sub;             // row ranges per half
};
inline Shape shape_of(int fout, int fin) {
    Shape s;
    s.TO = fout > 64 ? 2 : 1;
    s.TK = fin > 64 ? 2 : 1;
    s.A1 = fout > 32;
    s.B1 = fin > 32;
    s.sub = 4 / (s.TO * s.TK);
    return s;
}

inline int check(const void *Gm, int64_t ldg, const void *X, int64_t ldx, int64_t n, int fout, int fin, const void *dW, int64_t lddw,
                 const void *ws, int64_t ws_elems, int64_t need) {
    if (n < 0 || fout <= 0 || fin <= 0 || !dW || (n > 0 && (!Gm || !X))) return fail(-1, "pgcn_linear_weight_grad_f32: bad argument");
    if (fout > kMaxF || fin > kMaxF) return fail(-2, "pgcn_linear_weight_grad_f32: widths above 128 are left to the library GEMM");
    if (ldg < fout || ldx < fin || lddw < fin) return fail(-1, "pgcn_linear_weight_grad_f32: leading dimension below the width");
    if (16 * ldg * 4 >= ((int64_t)1 << 31) || 16 * ldx * 4 >= ((int64_t)1 << 31))
        return fail(-2, "pgcn_linear_weight_grad_f32: leading dimension too large for 32-bit lane offsets");
    if (n > ((int64_t)1 << 40)) return fail(-1, "pgcn_linear_weight_grad_f32: n out of range");
    if (!ws || ws_elems < need) return fail(-1, "pgcn_linear_weight_grad_f32: work-space too small");
    return 0;
}

inline int workgroups_for(int64_t n, int cus) {
    const int64_t steps = (n + kStepRows - 1) / kStepRows;
    // a half-wave set wants at least ~4 steps; never more workgroups than CUs (one partial matrix per workgroup and sub range)
    int64_t w = steps / 32;
    if (w < 1) w = 1;
    return (int)(w < cus ? w : cus);
}

}  // namespace pgcn_wgrad

extern "C" const char *pgcn_wgrad_last_error(void) { return pgcn_wgrad::g_err; }

// floats of work-space that any call needs at most (one 128 x 128 partial matrix per workgroup, at most 1024 workgroups / sub ranges)
extern "C" int64_t pgcn_linear_weight_grad_ws_elems(void) { return (int64_t)1024 * 128 * 128; }

namespace pgcn_wgrad {
template <bool F32>
int run(const float *Gm, int64_t ldg, const float *X, int64_t ldx, int64_t n, int32_t fout, int32_t fin, float *dW, int64_t lddw,
        float *ws, int64_t ws_elems, void *stream) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return fail(-1, "hipDeviceGetAttribute(MultiprocessorCount)");
    if (cus > 1024) cus = 1024;
    const Shape sh = shape_of(fout > 0 ? fout : 1, fin > 0 ? fin : 1);
    const int wgs = workgroups_for(n, cus);
    const int parts = wgs * sh.sub, FO = 64 * sh.TO, FI = 64 * sh.TK;
    if (int rc = check(Gm, ldg, X, ldx, n, fout, fin, dW, lddw, ws, ws_elems, (int64_t)parts * FO * FI)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {                                  // an empty sum
        if (hipMemset2DAsync(dW, (size_t)lddw * 4, 0, (size_t)fin * 4, (size_t)fout, s) != hipSuccess) return fail(-1, "hipMemset2DAsync");
        return 0;
    }
    // widths that do not fill the blocks of the instance: lanes beyond a width read a clamped column and drop it
    const bool ragged = fout % 32 != 0 || fin % 32 != 0 || (sh.A1 && fout <= 32 * (2 * sh.TO - 1)) || (sh.B1 && fin <= 32 * (2 * sh.TK - 1));
    {
        const int64_t steps = (n + kStepRows - 1) / kStepRows, per = steps / ((int64_t)wgs * 2 * sh.sub) + 2;
        if (per * kStepRows * (ldg > ldx ? ldg : ldx) * 4 >= ((int64_t)1 << 31))
            return fail(-2, "pgcn_linear_weight_grad_f32: a row range exceeds the 32-bit range of a buffer descriptor");
    }
#define PGCN_WGRAD_CASE(TO_, TK_, A1_, B1_)                                                                                   \
    if (sh.TO == TO_ && sh.TK == TK_ && sh.A1 == A1_ && sh.B1 == B1_) {                                                        \
        if (ragged) hipLaunchKernelGGL((wgrad_kernel<TO_, TK_, A1_, B1_, F32, true>), dim3((unsigned)wgs), dim3(kThreads), 0, s, Gm, ldg, X, \
                           ldx, n, fout, fin, ws);                                                                           \
        else hipLaunchKernelGGL((wgrad_kernel<TO_, TK_, A1_, B1_, F32, false>), dim3((unsigned)wgs), dim3(kThreads), 0, s, Gm, ldg, X,    \
                           ldx, n, fout, fin, ws);                                                                                    \
    } else
    PGCN_WGRAD_CASE(2, 2, true, true)
    PGCN_WGRAD_CASE(2, 1, true, true)
    PGCN_WGRAD_CASE(1, 2, true, true)
    PGCN_WGRAD_CASE(1, 1, true, true)
    PGCN_WGRAD_CASE(2, 1, true, false)
    PGCN_WGRAD_CASE(1, 2, false, true)
    PGCN_WGRAD_CASE(1, 1, true, false)
    PGCN_WGRAD_CASE(1, 1, false, true)
    PGCN_WGRAD_CASE(1, 1, false, false)
    { return fail(-2, "pgcn_linear_weight_grad_f32: no instance for these widths"); }
#undef PGCN_WGRAD_CASE
    if (hipGetLastError() != hipSuccess) return fail(-1, "kernel launch (wgrad_kernel)");
    const int chunks = FO * FI / 4;
    hipLaunchKernelGGL(wgrad_sum_kernel, dim3((unsigned)((chunks + 15) / 16)), dim3(256), 0, s, ws, parts, FO, FI, fout, fin, dW, lddw);
    return hipGetLastError() == hipSuccess ? 0 : fail(-1, "kernel launch (wgrad_sum_kernel)");
}
}  // namespace pgcn_wgrad

// dW (fout x fin, lddw) = Gm^T . X;  Gm: n x fout (ldg), X: n x fin (ldx), fp32 row-major;  ws: work-space of ws_elems floats
// (at least pgcn_linear_weight_grad_ws_elems()).  0; -2: shapes outside what the kernel takes (nothing launched: the caller runs the
// library product); -1: errors (pgcn_wgrad_last_error()).  Never allocates, never synchronises: two launches on `stream`.
extern "C" int pgcn_linear_weight_grad_f32(const float *Gm, int64_t ldg, const float *X, int64_t ldx, int64_t n, int32_t fout,
                                           int32_t fin, float *dW, int64_t lddw, float *ws, int64_t ws_elems, void *stream) {
    return pgcn_wgrad::run<PGCN_WGRAD_F32MFMA != 0>(Gm, ldg, X, ldx, n, fout, fin, dW, lddw, ws, ws_elems, stream);
}
#ifdef PGCN_WGRAD_PROBES
// measurement build (tools/micro/dense_fused_bench): the same kernel on the fp32 matrix cores
extern "C" int pgcn_linear_weight_grad_f32mfma_f32(const float *Gm, int64_t ldg, const float *X, int64_t ldx, int64_t n, int32_t fout,
                                                   int32_t fin, float *dW, int64_t lddw, float *ws, int64_t ws_elems, void *stream) {
    return pgcn_wgrad::run<true>(Gm, ldg, X, ldx, n, fout, fin, dW, lddw, ws, ws_elems, stream);
}
#endif
