// pgcn_loss.hip -- row-wise negative log-likelihood of a log-softmax, forward and backward, for gfx950.
//
//   loss_i = logsumexp_j x_ij - x_i,label_i            replaces F.nll_loss(F.log_softmax(logits, 1), labels)
//   dx_ij  = g * (exp(x_ij - lse_i) - [j == label_i])     /root/reference/GPU/PGCN.py:214-215 and its autograd graph
//
// The framework composes this from ~12 element-wise / reduction launches over the n x f logits (0.6 ms of a
// 13 ms epoch at the benchmark size); here it is one pass each way: a wave owns a row (f <= 64 * 16 = 1024
// columns, lane j takes columns j, j + 64, ...), maximum and sum by butterfly shuffles, fixed order =>
// bit-reproducible.  HBM-bound streams: 4 f bytes per row forward, 8 f bytes per row backward.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pgcn_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPerLane = 16;

__device__ __forceinline__ float wsum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(kThreads) void nll_rows_kernel(const float *__restrict__ X, int64_t ldx,
                                                            const int64_t *__restrict__ labels, int64_t nrows, int32_t f,
                                                            float *__restrict__ loss, float *__restrict__ lse) {
    const int64_t i = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nrows) return;
    const float *x = X + i * ldx;
    float v[kMaxPerLane];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) {
        const int j = lane + 64 * q;
        v[q] = j < f ? x[j] : -INFINITY;
        m = fmaxf(m, v[q]);
    }
    m = wmax(m);
    const float mm = isinf(m) ? 0.f : m;               // like torch.logsumexp: an all -inf (or +inf) row must not make NaN
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) s += (lane + 64 * q < f) ? expf(v[q] - mm) : 0.f;
    s = wsum(s);
    const float l = logf(s) + mm;
    if (lane == 0) {
        const int64_t y = labels[i];
        lse[i] = l;
        // a label outside [0, f) poisons the row's loss with NaN (F.nll_loss raises; no out-of-bounds read here)
        loss[i] = (y >= 0 && y < f) ? l - x[y] : __builtin_nanf("");
    }
}

__global__ __launch_bounds__(kThreads) void nll_rows_backward_kernel(const float *__restrict__ X, int64_t ldx,
                                                                     const int64_t *__restrict__ labels,
                                                                     const float *__restrict__ lse,
                                                                     const float *__restrict__ gscale, float scale,
                                                                     int64_t nrows, int32_t f, float *__restrict__ dX,
                                                                     int64_t lddx) {
    const int64_t i = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nrows) return;
    const float g = (gscale ? gscale[0] : 1.f) * scale;
    const float l = lse[i];
    const int64_t y = labels[i];
    const float *x = X + i * ldx;
    float *dx = dX + i * lddx;
    for (int j = lane; j < f; j += 64) dx[j] = g * (expf(x[j] - l) - (j == y ? 1.f : 0.f));
}

// f % 4 == 0, f <= 256, 16-byte aligned rows (the shapes of the training loops): 16 lanes own a row (float4 chunks
// c = sub, sub + 16, ...), four rows per wave -- a wave keeps 4 x more bytes in flight than with a row per wave (the
// kernel is a latency chain per row: 114 -> see DESIGN 6 at the benchmark size).  Fixed order => bit-reproducible.
constexpr int kV4Chunks = 4;

__device__ __forceinline__ float gsum16(float v) {
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gmax16(float v) {
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(kThreads) void nll_rows_v4_kernel(const float *__restrict__ X, int64_t ldx,
                                                               const int64_t *__restrict__ labels, int64_t nrows, int32_t f,
                                                               float *__restrict__ loss, float *__restrict__ lse) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int64_t i = ((int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool act = i < nrows;
    const float4 *x4 = reinterpret_cast<const float4 *>(X + (act ? i : 0) * ldx);
    const int nch = f >> 2;
    float4 v[kV4Chunks];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < kV4Chunks; ++q) {
        const int c = sub + 16 * q;
        if (act && c < nch) {
            v[q] = x4[c];
            m = fmaxf(fmaxf(fmaxf(m, v[q].x), fmaxf(v[q].y, v[q].z)), v[q].w);
        } else {
            v[q] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
    }
    m = gmax16(m);
    const float mm = isinf(m) ? 0.f : m;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < kV4Chunks; ++q)
        if (sub + 16 * q < nch) s += (expf(v[q].x - mm) + expf(v[q].y - mm)) + (expf(v[q].z - mm) + expf(v[q].w - mm));
    s = gsum16(s);
    const float l = logf(s) + mm;
    if (act && sub == 0) {
        const int64_t y = labels[i];
        lse[i] = l;
        loss[i] = (y >= 0 && y < f) ? l - X[i * ldx + y] : __builtin_nanf("");
    }
}

__global__ __launch_bounds__(kThreads) void nll_rows_backward_v4_kernel(const float *__restrict__ X, int64_t ldx,
                                                                        const int64_t *__restrict__ labels,
                                                                        const float *__restrict__ lse,
                                                                        const float *__restrict__ gscale, float scale,
                                                                        int64_t nrows, int32_t f, float *__restrict__ dX,
                                                                        int64_t lddx) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int64_t i = ((int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    if (i >= nrows) return;
    const float g = (gscale ? gscale[0] : 1.f) * scale;
    const float l = lse[i];
    const int y = (int)labels[i];
    const float4 *x4 = reinterpret_cast<const float4 *>(X + i * ldx);
    float4 *d4 = reinterpret_cast<float4 *>(dX + i * lddx);
    const int nch = f >> 2;
    for (int c = sub; c < nch; c += 16) {
        const float4 x = x4[c];
        const int j = 4 * c;
        d4[c] = make_float4(g * (expf(x.x - l) - (j == y ? 1.f : 0.f)), g * (expf(x.y - l) - (j + 1 == y ? 1.f : 0.f)),
                            g * (expf(x.z - l) - (j + 2 == y ? 1.f : 0.f)), g * (expf(x.w - l) - (j + 3 == y ? 1.f : 0.f)));
    }
}

}  // namespace

extern "C" int pgcn_nll_rows_f32(const float *X, int64_t ldx, const int64_t *labels, int64_t nrows, int32_t f,
                                 float *loss_rows, float *lse_rows, pgcn_stream_t stream) {
    if (nrows < 0 || f <= 0 || ldx < f) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_f32: bad sizes");
    if (f > 64 * kMaxPerLane) return pgcn_set_error(PGCN_EUNSUPPORTED, "pgcn_nll_rows_f32: more than 1024 columns");
    if (nrows == 0) return PGCN_OK;
    if (!X || !labels || !loss_rows || !lse_rows) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_f32: null pointer");
    if (f % 4 == 0 && f <= 64 * kV4Chunks && ldx % 4 == 0 && (uintptr_t)X % 16 == 0) {
        const int64_t g4 = (nrows + 15) / 16;
        if (g4 > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_f32: too many rows");
        hipLaunchKernelGGL(nll_rows_v4_kernel, dim3((unsigned)g4), dim3(kThreads), 0, (hipStream_t)stream, X, ldx, labels, nrows,
                           f, loss_rows, lse_rows);
        PGCN_HIP_CHECK(hipGetLastError());
        return PGCN_OK;
    }
    const int64_t grid = (nrows + 3) / 4;
    if (grid > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_f32: too many rows");
    hipLaunchKernelGGL(nll_rows_kernel, dim3((unsigned)grid), dim3(kThreads), 0, (hipStream_t)stream, X, ldx, labels, nrows,
                       f, loss_rows, lse_rows);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

extern "C" int pgcn_nll_rows_backward_f32(const float *X, int64_t ldx, const int64_t *labels, const float *lse_rows,
                                          const float *gscale_dev, float scale, int64_t nrows, int32_t f, float *dX,
                                          int64_t lddx, pgcn_stream_t stream) {
    if (nrows < 0 || f <= 0 || ldx < f || lddx < f) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_backward_f32: bad sizes");
    if (nrows == 0) return PGCN_OK;
    if (!X || !labels || !lse_rows || !dX) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_backward_f32: null pointer");
    if (f % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && (uintptr_t)X % 16 == 0 && (uintptr_t)dX % 16 == 0) {
        const int64_t g4 = (nrows + 15) / 16;
        if (g4 > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_backward_f32: too many rows");
        hipLaunchKernelGGL(nll_rows_backward_v4_kernel, dim3((unsigned)g4), dim3(kThreads), 0, (hipStream_t)stream, X, ldx,
                           labels, lse_rows, gscale_dev, scale, nrows, f, dX, lddx);
        PGCN_HIP_CHECK(hipGetLastError());
        return PGCN_OK;
    }
    const int64_t grid = (nrows + 3) / 4;
    if (grid > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_nll_rows_backward_f32: too many rows");
    hipLaunchKernelGGL(nll_rows_backward_kernel, dim3((unsigned)grid), dim3(kThreads), 0, (hipStream_t)stream, X, ldx, labels,
                       lse_rows, gscale_dev, scale, nrows, f, dX, lddx);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
