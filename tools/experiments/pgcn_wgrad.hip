// pgcn_wgrad.hip -- the weight gradient of a layer, dW = Gm^T . X (fout x fin, summed over the n rows), on the bf16 matrix
// cores at fp32 accuracy: the third dense product of `F.relu(self.linear(AH))` (/root/reference/GPU/PGCN.py:146-147; autograd's
// `grad_output.t() @ input`).  Rounds 1-4: a 64-slab batched library GEMM + a 64-way sum (PGCN._LinearNoBias.weight_grad, 78 us
// + the sum at n = 232 965, f = 128); the plain `g.t() @ x` runs on 16 workgroups (432 us).
//
// NOT YET RUN ON HARDWARE (written after the r04 GPU budget was spent): checked through the host build of this file only
// (tests/test_zz_dense_fused.py).  tools/micro/dense_fused_bench times it beside the other two products in r05.
//
// The contraction runs over ROWS: for v_mfma_f32_32x32x16_bf16 with M = 32 columns o of Gm, N = 32 columns k of X and the MFMA's
// k dimension = 16 rows, a lane (lo, hi) needs Gm[row0 + 8 hi + j][o0 + lo] and X[row0 + 8 hi + j][k0 + lo], j = 0..7: one
// column, eight consecutive rows -- dword loads that are coalesced ACROSS lanes (32 consecutive columns = one 128-byte line per
// row), every byte of both matrices used.  One workgroup per CU owns a contiguous range of rows; wave w = (column block
// ob = w & 3 of Gm, half = w >> 2): it multiplies the 16-row steps `half` (mod 2) of the range for its 32 columns of Gm against
// ALL column blocks of X (one A operand, up to four B operands, 24 MFMAs per step) into 4 x 16 accumulator registers, keeps the
// next step's 40 values in flight under them, and at the end the two halves are added through LDS and the workgroup writes ONE
// partial fout x fin matrix; a second kernel sums the partials in a fixed order (deterministic, no atomics).
// Split: the three-plane bf16 split of pgcn_dense_common.h, the same six partial products.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#ifdef PGCN_DENSE_HOST_EMU
#define PG_HD inline
#else
#include <hip/hip_runtime.h>
#define PG_HD __device__ __forceinline__
#endif

namespace pgcn_wgrad {

#include "pgcn_dense_common.h"

constexpr int kThreads = 512;
constexpr int kStepRows = 16;             // rows of one MFMA step (its k dimension)
constexpr int kMaxF = 128;
constexpr int kMaxParts = 512;            // workgroups = partial matrices (one per CU)

// the eight values of lane (lo, hi) for one operand of one step: M[row0 + 8 hi + j][col0 + lo]; zeros outside rows < row_end, width
PG_HD void load_column8(float (&v)[8], const float *__restrict__ M, int64_t ld, int64_t row0, int64_t row_end, int col0, int width,
                        int lane) {
    const int col = col0 + (lane & 31);
    const int64_t r = row0 + 8 * (lane >> 5);
    const int colc = col < width ? col : width - 1;
    // clamped addresses, values dropped afterwards (no branches); one multiply, then conditional increments of one row
    const int64_t rmax = row_end - 1;
    int64_t off = (r < rmax ? r : rmax) * ld + colc;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = M[off];
        v[j] = (r + j < row_end && col < width) ? x : 0.f;
        off += (r + j < rmax) ? ld : 0;
    }
}
PG_HD void planes_of(const float (&v)[8], u32x4 (&p)[3]) {
    const f32x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
    split8(lo4, hi4, p);
}
// rows [begin, end) of workgroup `part` of `parts`: contiguous, multiples of 32 rows (both halves get whole steps)
PG_HD void rows_of_part(int64_t n, int parts, int part, int64_t &begin, int64_t &end) {
    const int64_t per = ((n + parts - 1) / parts + 31) / 32 * 32;
    begin = per * part < n ? per * part : n;
    end = begin + per < n ? begin + per : n;
}
// accumulator register r of lane (lo, hi), block kb of wave column block ob -> element (32 ob + (r & 3) + 8 (r >> 2) + 4 hi, 32 kb + lo)
PG_HD int partial_index(int ob, int kb, int r, int lane, int ldp) {
    return (32 * ob + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * ldp + 32 * kb + (lane & 31);
}

thread_local char g_err[256] = "";
int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

#ifndef PGCN_DENSE_HOST_EMU
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NKB>
struct StepVals {
    float a[8], b[NKB][8];
};
template <int NKB>
PG_HD void load_step(StepVals<NKB> &s, const float *__restrict__ Gm, int64_t ldg, const float *__restrict__ X, int64_t ldx,
                     int64_t row0, int64_t row_end, int ob, int fout, int fin, int lane) {
    load_column8(s.a, Gm, ldg, row0, row_end, 32 * ob, fout, lane);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) load_column8(s.b[kb], X, ldx, row0, row_end, 32 * kb, fin, lane);
}
template <int NKB>
PG_HD void step_product(const StepVals<NKB> &s, f32x16 (&acc)[NKB]) {
    u32x4 a[3];
    planes_of(s.a, a);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        PGCN_DENSE_PRODUCTS;
        u32x4 b[3];
        planes_of(s.b[kb], b);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[kPA[i]]), __builtin_bit_cast(bf16x8, b[kPB[i]]),
                                                               acc[kb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// partial[part] (32 NOB x ldp, ldp = 32 NKB) = Gm[rows of part]^T . X[rows of part]; NOB = column blocks of Gm (waves ob >= NOB idle)
template <int NKB>
__global__ __launch_bounds__(kThreads, 2) void wgrad_kernel(const float *__restrict__ Gm, int64_t ldg, const float *__restrict__ X,
                                                            int64_t ldx, int64_t n, int fout, int fin, int nob,
                                                            float *__restrict__ partial) {
    __shared__ float other[4 * NKB * 16 * 64];                  // the accumulators of the waves of half 1: [ob][kb][r][lane]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ob = w & 3, half = w >> 2;
    int64_t begin, end;
    rows_of_part(n, gridDim.x, blockIdx.x, begin, end);
    f32x16 acc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
    if (ob < nob) {
        StepVals<NKB> s0, s1;
        int64_t row0 = begin + kStepRows * half;
        if (row0 < end) load_step<NKB>(s0, Gm, ldg, X, ldx, row0, end, ob, fout, fin, lane);
        while (row0 < end) {                                      // two register sets that swap roles
            int64_t rn = row0 + 2 * kStepRows;
            if (rn < end) load_step<NKB>(s1, Gm, ldg, X, ldx, rn, end, ob, fout, fin, lane);
            __builtin_amdgcn_sched_barrier(0);
            step_product<NKB>(s0, acc);
            row0 = rn;
            if (row0 >= end) break;
            rn = row0 + 2 * kStepRows;
            if (rn < end) load_step<NKB>(s0, Gm, ldg, X, ldx, rn, end, ob, fout, fin, lane);
            __builtin_amdgcn_sched_barrier(0);
            step_product<NKB>(s1, acc);
            row0 = rn;
        }
    }
    if (half == 1) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) other[((ob * NKB + kb) * 16 + r) * 64 + lane] = acc[kb][r];
    }
    __syncthreads();
    if (half == 0 && ob < nob) {
        float *out = partial + (int64_t)blockIdx.x * (32 * nob) * (32 * NKB);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[partial_index(ob, kb, r, lane, 32 * NKB)] = acc[kb][r] + other[((ob * NKB + kb) * 16 + r) * 64 + lane];
    }
}

// dW[o][k] = sum over parts, in order, of partial[part][o][k]
__global__ __launch_bounds__(256) void wgrad_sum_kernel(const float *__restrict__ partial, int parts, int rows_p, int ldp, int fout,
                                                        int fin, float *__restrict__ dW, int64_t lddw) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= fout * fin) return;
    const int o = e / fin, k = e % fin;
    const float *src = partial + (int64_t)o * ldp + k;
    const int64_t slab = (int64_t)rows_p * ldp;
    float s = 0.f;
    for (int p = 0; p < parts; ++p) s += src[p * slab];
    dW[o * lddw + k] = s;
}
#endif

int check(const void *Gm, int64_t ldg, const void *X, int64_t ldx, int64_t n, int fout, int fin, const void *dW, int64_t lddw,
          const void *ws, int64_t ws_elems, int parts) {
    if (n < 0 || fout <= 0 || fin <= 0 || !dW || (n > 0 && (!Gm || !X))) return fail(-1, "pgcn_linear_weight_grad_f32: bad argument");
    if (fout > kMaxF || fin > kMaxF) return fail(-2, "pgcn_linear_weight_grad_f32: widths above 128 are left to the library GEMM");
    if (ldg < fout || ldx < fin || lddw < fin) return fail(-1, "pgcn_linear_weight_grad_f32: leading dimension below the width");
    const int nob = (fout + 31) / 32, nkb = (fin + 31) / 32;
    if (!ws || ws_elems < (int64_t)parts * 32 * nob * 32 * nkb) return fail(-1, "pgcn_linear_weight_grad_f32: work-space too small");
    return 0;
}

}  // namespace pgcn_wgrad

extern "C" const char *pgcn_wgrad_last_error(void) { return pgcn_wgrad::g_err; }

// floats of work-space that any call needs at most (one partial 128 x 128 matrix per workgroup, at most 512 workgroups)
extern "C" int64_t pgcn_linear_weight_grad_ws_elems(void) { return (int64_t)pgcn_wgrad::kMaxParts * 128 * 128; }

#ifndef PGCN_DENSE_HOST_EMU
// dW (fout x fin, lddw) = Gm^T . X;  Gm: n x fout (ldg), X: n x fin (ldx);  ws: work-space of ws_elems floats.
// 0; -2: widths outside what the kernel takes (nothing launched); -1: errors (pgcn_wgrad_last_error()).
extern "C" int pgcn_linear_weight_grad_f32(const float *Gm, int64_t ldg, const float *X, int64_t ldx, int64_t n, int32_t fout,
                                           int32_t fin, float *dW, int64_t lddw, float *ws, int64_t ws_elems, void *stream) {
    using namespace pgcn_wgrad;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
        return fail(-1, "hipDeviceGetAttribute(MultiprocessorCount)");
    const int64_t steps = (n + 31) / 32;
    int parts = (int)(steps < cus ? (steps > 0 ? steps : 1) : cus);
    if (parts > kMaxParts) parts = kMaxParts;
    if (int rc = check(Gm, ldg, X, ldx, n, fout, fin, dW, lddw, ws, ws_elems, parts)) return rc;
    const int nob = (fout + 31) / 32, nkb = (fin + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
    switch (nkb) {
        case 1: hipLaunchKernelGGL(wgrad_kernel<1>, dim3(parts), dim3(kThreads), 0, s, Gm, ldg, X, ldx, n, fout, fin, nob, ws); break;
        case 2: hipLaunchKernelGGL(wgrad_kernel<2>, dim3(parts), dim3(kThreads), 0, s, Gm, ldg, X, ldx, n, fout, fin, nob, ws); break;
        case 3: hipLaunchKernelGGL(wgrad_kernel<3>, dim3(parts), dim3(kThreads), 0, s, Gm, ldg, X, ldx, n, fout, fin, nob, ws); break;
        default: hipLaunchKernelGGL(wgrad_kernel<4>, dim3(parts), dim3(kThreads), 0, s, Gm, ldg, X, ldx, n, fout, fin, nob, ws); break;
    }
    if (hipGetLastError() != hipSuccess) return fail(-1, "wgrad_kernel launch");
    hipLaunchKernelGGL(wgrad_sum_kernel, dim3((fout * fin + 255) / 256), dim3(256), 0, s, ws, parts, 32 * nob, 32 * nkb, fout, fin, dW, lddw);
    return hipGetLastError() == hipSuccess ? 0 : fail(-1, "wgrad_sum_kernel launch");
}

#else
// ---- host emulation: the kernel's loaders, partition of the rows and partial layout around an emulated MFMA ----------------------
namespace {
using namespace pgcn_wgrad;

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
}  // namespace

// the same entry point on HOST pointers; `stream` carries the number of workgroups to emulate (0: 3)
extern "C" int pgcn_linear_weight_grad_f32(const float *Gm, int64_t ldg, const float *X, int64_t ldx, int64_t n, int32_t fout,
                                           int32_t fin, float *dW, int64_t lddw, float *ws, int64_t ws_elems, void *stream) {
    using namespace pgcn_wgrad;
    int parts = (int)(intptr_t)stream;
    if (parts <= 0) parts = 3;
    if (int rc = check(Gm, ldg, X, ldx, n, fout, fin, dW, lddw, ws, ws_elems, parts)) return rc;
    const int nob = (fout + 31) / 32, nkb = (fin + 31) / 32, ldp = 32 * nkb;
    for (int part = 0; part < parts; ++part) {
        int64_t begin, end;
        rows_of_part(n, parts, part, begin, end);
        float *out = ws + (int64_t)part * (32 * nob) * ldp;
        for (int ob = 0; ob < nob; ++ob) {
            static f32x16 acc[2][4][64];
            for (int half = 0; half < 2; ++half) {
                for (int kb = 0; kb < nkb; ++kb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 16; ++r) acc[half][kb][lane][r] = 0.f;
                for (int64_t row0 = begin + kStepRows * half; row0 < end; row0 += 2 * kStepRows) {
                    static u32x4 a[3][64], b[3][64];
                    for (int lane = 0; lane < 64; ++lane) {
                        float v[8];
                        u32x4 p[3];
                        load_column8(v, Gm, ldg, row0, end, 32 * ob, fout, lane);
                        planes_of(v, p);
                        for (int pl = 0; pl < 3; ++pl) a[pl][lane] = p[pl];
                    }
                    for (int kb = 0; kb < nkb; ++kb) {
                        for (int lane = 0; lane < 64; ++lane) {
                            float v[8];
                            u32x4 p[3];
                            load_column8(v, X, ldx, row0, end, 32 * kb, fin, lane);
                            planes_of(v, p);
                            for (int pl = 0; pl < 3; ++pl) b[pl][lane] = p[pl];
                        }
                        PGCN_DENSE_PRODUCTS;
                        for (int i = 0; i < 6; ++i) mfma_emu(a[kPA[i]], b[kPB[i]], acc[half][kb]);
                    }
                }
            }
            for (int kb = 0; kb < nkb; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 16; ++r) out[partial_index(ob, kb, r, lane, ldp)] = acc[0][kb][lane][r] + acc[1][kb][lane][r];
        }
    }
    for (int o = 0; o < fout; ++o)
        for (int k = 0; k < fin; ++k) {
            float s = 0.f;
            for (int p = 0; p < parts; ++p) s += ws[(int64_t)p * (32 * nob) * ldp + (int64_t)o * ldp + k];
            dW[o * lddw + k] = s;
        }
    return 0;
}
#endif
