// pgcn_gat.hip -- attention kernels of the GAT path for gfx950.
//
// Replaces the dense n x n arithmetic of PGAT.forward (GPU/PGAT.py:138-151) by work on the stored
// entries only.  Per layer and head k, with s1 = Z a1, s2 = Z a2 (PGAT.py:141-142):
//
//   edge softmax   raw_ij = s1_i + s2_j                                    (PGAT.py:144)
//     mode 0 (standard GAT)   e = LeakyReLU(raw), alpha_ij = softmax over the entries of row i
//     mode 1 (reference)      every one of the n_global columns takes part, non-edges with logit 0
//                             (PGAT.py:145-147): m = max(0, max e), D = sum_edges exp(e-m) + (n-deg) exp(-m),
//                             alpha_ij = (exp(e_ij-m) - exp(-m)) / D,  beta_i = exp(-m) / D,
//                             so that out_i = sum_edges alpha_ij Z_j + beta_i sum_all Z_j          (PGAT.py:149)
//   edge gradient  dp_ij = <dOut_i, Z_j>,  de_ij = (alpha_ij + beta_i)(dp_ij - t_i) [x LeakyReLU'],
//                  t_i = <dOut_i, out_i>;  ds1_i = sum_j de_ij   (ds2_j = sum_i de_ij: pgcn_csr_row_sums_f32
//                  on the transposed structure; dZ = alpha^T dOut: the SpMM kernels with permuted values)
//
// alpha / de are stored head-major, [heads][nnz], in the storage order of `col`: head k's plane is
// the `val` array of the CSR SpMM kernels.  The aggregation itself is pgcn_spmm_csr_plan_f32.
//
// Rooflines.  Softmax: HBM stream, 4 B (col) + 4 B (alpha) per entry and head; s2 (n x heads fp32) is
// L2 resident.  Edge gradient: one gather of d floats of Z per entry and head -- the same L2-bound
// gather as the SpMM -- plus 8 B of streams.  One 64-lane wave works on one (row, head); rows
// longer than the host's threshold get a 256-thread workgroup instead (hub rows of power-law graphs).
// No atomics: every output element has one writer, sums run in a fixed order => bit-reproducible.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pgcn_internal.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float wave_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// TPR = 64: the caller's wave; TPR = 256: the whole workgroup (4 waves through LDS)
template <int TPR, bool MAX>
__device__ __forceinline__ float group_reduce(float v, float *red) {
    v = MAX ? wave_max(v) : wave_sum(v, 64);
    if constexpr (TPR == 64) return v;
    __syncthreads();                       // red[] may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : ((a + b) + (c + d));
}

template <int TPR>
__device__ __forceinline__ bool pick_row(const int32_t *rows, int64_t nlist, int64_t &i, int &lane) {
    int64_t li;
    if constexpr (TPR == 64) {
        li = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
        lane = threadIdx.x & 63;
    } else {
        li = blockIdx.x;
        lane = threadIdx.x;
    }
    if (li >= nlist) return false;
    i = rows ? (int64_t)rows[li] : li;
    return true;
}

template <int TPR, int MODE>
__global__ __launch_bounds__(kThreads) void gat_softmax_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ s1, int64_t lds1, const float *__restrict__ s2, int64_t lds2,
    int32_t heads, float slope, float nglobal, float *__restrict__ alpha, float *__restrict__ beta, int64_t nnz) {
    __shared__ float red[4];
    int64_t i;
    int lane;
    if (!pick_row<TPR>(rows, nlist, i, lane)) return;
    const int k = blockIdx.y;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    const float a = s1[i * lds1 + k];
    const float *s2k = s2 + k;
    auto score = [&](int64_t p) {
        float r = a + s2k[(int64_t)col[p] * lds2];
        if (MODE == 0) r = r > 0.f ? r : r * slope;
        return r;
    };
    float m = MODE == 1 ? 0.f : -INFINITY;
    for (int64_t p = b + lane; p < e; p += TPR) m = fmaxf(m, score(p));
    m = group_reduce<TPR, true>(m, red);
    float sum = 0.f;
    for (int64_t p = b + lane; p < e; p += TPR) sum += expf(score(p) - m);
    sum = group_reduce<TPR, false>(sum, red);
    float em = 0.f, D = sum;
    if (MODE == 1) {
        em = expf(-m);
        D = (nglobal - (float)(e - b)) * em + sum;
    }
    const float inv = D > 0.f ? 1.f / D : 0.f;
    float *ak = alpha + (int64_t)k * nnz;
    for (int64_t p = b + lane; p < e; p += TPR) ak[p] = (expf(score(p) - m) - em) * inv;
    if (MODE == 1 && lane == 0) beta[i * heads + k] = em * inv;
}

template <int VEC> struct Vec;
template <> struct Vec<1> {
    using T = float;
    static __device__ __forceinline__ float dot(float a, float b) { return a * b; }
};
template <> struct Vec<4> {
    using T = float4;
    static __device__ __forceinline__ float dot(float4 a, float4 b) {
        return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
};

template <int TPR, int VEC, int MODE>
__global__ __launch_bounds__(kThreads) void gat_edge_grad_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ s1, int64_t lds1, const float *__restrict__ s2, int64_t lds2,
    const float *__restrict__ alpha, const float *__restrict__ beta, const float *__restrict__ Z, int64_t ldz,
    const float *__restrict__ dOut, int64_t ldo, const float *__restrict__ t, int32_t heads, int32_t d, int32_t lpe,
    float slope, float *__restrict__ de, float *__restrict__ ds1, int64_t nnz) {
    using V = typename Vec<VEC>::T;
    __shared__ float red[4];
    int64_t i;
    int lane;
    if (!pick_row<TPR>(rows, nlist, i, lane)) return;
    const int k = blockIdx.y;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    const int nvec = d / VEC;
    const int sub = lane & (lpe - 1);      // lane inside the lpe-lane team that works on one entry
    const int team = lane / lpe;
    const int epi = TPR / lpe;             // entries per iteration
    const V *go = reinterpret_cast<const V *>(dOut + i * ldo + (int64_t)k * d);
    const float a = s1[i * lds1 + k];
    const float ti = t[i * heads + k];
    const float bi = MODE == 1 ? beta[i * heads + k] : 0.f;
    const float *ak = alpha + (int64_t)k * nnz;
    float *dk = de + (int64_t)k * nnz;
    const bool one = nvec <= lpe;          // the team covers the head in one vector per lane
    V g0 = {};
    if (one && sub < nvec) g0 = go[sub];
    float acc = 0.f;
    // two entries per team and iteration: their gathers and shuffle chains overlap
    for (int64_t p0 = b; p0 < e; p0 += 2 * epi) {
        const int64_t pa = p0 + team, pb = pa + epi;
        const bool va = pa < e, vb = pb < e;
        const int64_t ca = col[va ? pa : e - 1], cb = col[vb ? pb : e - 1];
        const V *za = reinterpret_cast<const V *>(Z + ca * ldz + (int64_t)k * d);
        const V *zb = reinterpret_cast<const V *>(Z + cb * ldz + (int64_t)k * d);
        float da = 0.f, db = 0.f;
        if (one) {
            if (sub < nvec) {
                const V xa = za[sub], xb = zb[sub];
                da = Vec<VEC>::dot(g0, xa);
                db = Vec<VEC>::dot(g0, xb);
            }
        } else {
            for (int v = sub; v < nvec; v += lpe) {
                const V gv = go[v];
                da += Vec<VEC>::dot(gv, za[v]);
                db += Vec<VEC>::dot(gv, zb[v]);
            }
        }
        for (int o = lpe >> 1; o > 0; o >>= 1) {
            da += __shfl_xor(da, o, 64);
            db += __shfl_xor(db, o, 64);
        }
        if (sub == 0) {
            float ga = 0.f, gb = 0.f;
            if (va) {
                ga = (ak[pa] + bi) * (da - ti);
                if (MODE == 0) ga *= (a + s2[ca * lds2 + k]) > 0.f ? 1.f : slope;
                dk[pa] = ga;
            }
            if (vb) {
                gb = (ak[pb] + bi) * (db - ti);
                if (MODE == 0) gb *= (a + s2[cb * lds2 + k]) > 0.f ? 1.f : slope;
                dk[pb] = gb;
            }
            acc += ga + gb;
        }
    }
    acc = group_reduce<TPR, false>(acc, red);
    if (lane == 0) ds1[i * heads + k] = acc;
}

template <int TPR>
__global__ __launch_bounds__(kThreads) void csr_row_sums_kernel(
    const int64_t *__restrict__ rowptr, const int64_t *__restrict__ perm, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ src, int64_t nnz, float *__restrict__ out, int64_t ldo) {
    __shared__ float red[4];
    int64_t i;
    int lane;
    if (!pick_row<TPR>(rows, nlist, i, lane)) return;
    const int k = blockIdx.y;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    const float *sk = src + (int64_t)k * nnz;
    float acc = 0.f;
    if (perm)
        for (int64_t p = b + lane; p < e; p += TPR) acc += sk[perm[p]];
    else
        for (int64_t p = b + lane; p < e; p += TPR) acc += sk[p];
    acc = group_reduce<TPR, false>(acc, red);
    if (lane == 0) out[i * ldo + k] = acc;
}

__global__ __launch_bounds__(kThreads) void csr_permute_kernel(const float *__restrict__ src,
                                                               const int64_t *__restrict__ perm, int64_t nnz,
                                                               float *__restrict__ dst) {
    const float *s = src + (int64_t)blockIdx.y * nnz;
    float *d = dst + (int64_t)blockIdx.y * nnz;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * kThreads)
        d[p] = s[perm[p]];
}

struct RowLists {
    const int32_t *wave;
    int64_t nwave;
    const int32_t *block;
    int64_t nblock;
};

int check_lists(const char *who, int64_t nrows, const RowLists &l) {
    if (l.nwave < 0 || l.nblock < 0 || l.nwave + l.nblock > nrows || (l.nblock && !l.block))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad row lists");
    if (l.nwave > 0x7fffffffLL * 4 || l.nblock > 0x7fffffffLL) return pgcn_set_error2(PGCN_EINVAL, who, "too many rows");
    return PGCN_OK;
}

inline dim3 wave_grid(int64_t n, int heads) { return dim3((unsigned)((n + 3) / 4), (unsigned)heads); }
inline dim3 block_grid(int64_t n, int heads) { return dim3((unsigned)n, (unsigned)heads); }

}  // namespace

extern "C" int pgcn_gat_edge_softmax_f32(const int64_t *rowptr, const int32_t *col, int64_t nrows, int64_t nnz,
                                         const int32_t *rows_wave, int64_t nrows_wave, const int32_t *rows_block,
                                         int64_t nrows_block, const float *s1, int64_t lds1, const float *s2,
                                         int64_t lds2, int32_t heads, float slope, int32_t mode, int64_t n_global,
                                         float *alpha, float *beta, pgcn_stream_t stream) {
    const char *who = "pgcn_gat_edge_softmax_f32";
    if (nrows < 0 || nnz < 0 || heads < 1 || heads > 65535 || lds1 < heads || lds2 < heads || (mode != 0 && mode != 1))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    const RowLists l{rows_wave, nrows_wave, rows_block, nrows_block};
    int rc = check_lists(who, nrows, l);
    if (rc != PGCN_OK) return rc;
    if (nrows == 0 || l.nwave + l.nblock == 0) return PGCN_OK;
    if (!rowptr || !s1 || (nnz && (!col || !s2 || !alpha)) || (mode == 1 && !beta))
        return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const float ng = (float)n_global;
#define PGCN_SOFTMAX(TPR, MODE, GRID, ROWS, N)                                                                   \
    hipLaunchKernelGGL((gat_softmax_kernel<TPR, MODE>), GRID, dim3(kThreads), 0, s, rowptr, col, ROWS, N, s1, lds1, \
                       s2, lds2, heads, slope, ng, alpha, beta, nnz)
    if (l.nwave) {
        if (mode == 0) PGCN_SOFTMAX(64, 0, wave_grid(l.nwave, heads), l.wave, l.nwave);
        else PGCN_SOFTMAX(64, 1, wave_grid(l.nwave, heads), l.wave, l.nwave);
    }
    if (l.nblock) {
        if (mode == 0) PGCN_SOFTMAX(256, 0, block_grid(l.nblock, heads), l.block, l.nblock);
        else PGCN_SOFTMAX(256, 1, block_grid(l.nblock, heads), l.block, l.nblock);
    }
#undef PGCN_SOFTMAX
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

extern "C" int pgcn_gat_edge_grad_f32(const int64_t *rowptr, const int32_t *col, int64_t nrows, int64_t nnz,
                                      const int32_t *rows_wave, int64_t nrows_wave, const int32_t *rows_block,
                                      int64_t nrows_block, const float *s1, int64_t lds1, const float *s2,
                                      int64_t lds2, const float *alpha, const float *beta, const float *Z,
                                      int64_t ldz, const float *dOut, int64_t ldo, const float *t, int32_t heads,
                                      int32_t d, float slope, int32_t mode, float *de, float *ds1,
                                      pgcn_stream_t stream) {
    const char *who = "pgcn_gat_edge_grad_f32";
    if (nrows < 0 || nnz < 0 || heads < 1 || heads > 65535 || d < 1 || lds1 < heads || lds2 < heads ||
        ldz < (int64_t)heads * d || ldo < (int64_t)heads * d || (mode != 0 && mode != 1))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    const RowLists l{rows_wave, nrows_wave, rows_block, nrows_block};
    int rc = check_lists(who, nrows, l);
    if (rc != PGCN_OK) return rc;
    if (nrows == 0 || l.nwave + l.nblock == 0) return PGCN_OK;
    if (!rowptr || !s1 || !t || !ds1 || !dOut || (nnz && (!col || !s2 || !alpha || !Z || !de)) || (mode == 1 && !beta))
        return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = d % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && (uintptr_t)Z % 16 == 0 && (uintptr_t)dOut % 16 == 0;
    const int nvec = v4 ? d / 4 : d;
    int lpe = 1;
    while (lpe < nvec && lpe < 64) lpe *= 2;
#define PGCN_EGRAD(TPR, VEC, MODE, GRID, ROWS, N)                                                                  \
    hipLaunchKernelGGL((gat_edge_grad_kernel<TPR, VEC, MODE>), GRID, dim3(kThreads), 0, s, rowptr, col, ROWS, N, s1, \
                       lds1, s2, lds2, alpha, beta, Z, ldz, dOut, ldo, t, heads, d, lpe, slope, de, ds1, nnz)
#define PGCN_EGRAD_VM(TPR, GRID, ROWS, N)                                  \
    do {                                                                   \
        if (v4 && mode == 0) PGCN_EGRAD(TPR, 4, 0, GRID, ROWS, N);         \
        else if (v4) PGCN_EGRAD(TPR, 4, 1, GRID, ROWS, N);                 \
        else if (mode == 0) PGCN_EGRAD(TPR, 1, 0, GRID, ROWS, N);          \
        else PGCN_EGRAD(TPR, 1, 1, GRID, ROWS, N);                         \
    } while (0)
    if (l.nwave) PGCN_EGRAD_VM(64, wave_grid(l.nwave, heads), l.wave, l.nwave);
    if (l.nblock) PGCN_EGRAD_VM(256, block_grid(l.nblock, heads), l.block, l.nblock);
#undef PGCN_EGRAD_VM
#undef PGCN_EGRAD
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

extern "C" int pgcn_csr_row_sums_f32(const int64_t *rowptr, const int64_t *perm, int64_t nrows, int64_t nnz,
                                     const int32_t *rows_wave, int64_t nrows_wave, const int32_t *rows_block,
                                     int64_t nrows_block, const float *src, int32_t planes, float *out, int64_t ldo,
                                     pgcn_stream_t stream) {
    const char *who = "pgcn_csr_row_sums_f32";
    if (nrows < 0 || nnz < 0 || planes < 1 || planes > 65535 || ldo < planes)
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    const RowLists l{rows_wave, nrows_wave, rows_block, nrows_block};
    int rc = check_lists(who, nrows, l);
    if (rc != PGCN_OK) return rc;
    if (nrows == 0 || l.nwave + l.nblock == 0) return PGCN_OK;
    if (!rowptr || !out || (nnz && !src)) return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (l.nwave)
        hipLaunchKernelGGL((csr_row_sums_kernel<64>), wave_grid(l.nwave, planes), dim3(kThreads), 0, s, rowptr, perm,
                           l.wave, l.nwave, src, nnz, out, ldo);
    if (l.nblock)
        hipLaunchKernelGGL((csr_row_sums_kernel<256>), block_grid(l.nblock, planes), dim3(kThreads), 0, s, rowptr, perm,
                           l.block, l.nblock, src, nnz, out, ldo);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

extern "C" int pgcn_csr_permute_f32(const float *src, const int64_t *perm, int64_t nnz, int32_t planes, float *dst,
                                    pgcn_stream_t stream) {
    if (nnz < 0 || planes < 1 || planes > 65535) return pgcn_set_error(PGCN_EINVAL, "pgcn_csr_permute_f32: bad sizes");
    if (nnz == 0) return PGCN_OK;
    if (!src || !perm || !dst) return pgcn_set_error(PGCN_EINVAL, "pgcn_csr_permute_f32: null pointer");
    int64_t grid = (nnz + kThreads - 1) / kThreads;
    if (grid > 256 * 32) grid = 256 * 32;
    hipLaunchKernelGGL(csr_permute_kernel, dim3((unsigned)grid, (unsigned)planes), dim3(kThreads), 0, (hipStream_t)stream,
                       src, perm, nnz, dst);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
