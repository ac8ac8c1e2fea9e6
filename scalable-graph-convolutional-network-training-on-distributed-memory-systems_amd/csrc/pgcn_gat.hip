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

// alpha of the TRANSPOSED structure, recomputed from the per-row softmax statistics instead of
// gathered through a permutation (a 4-byte random gather per entry from a GB-sized array):
// entry (row j, col i) of A^T is entry (i, j) of A, alpha = (exp(e - m_i) - em_i) * inv_i with
// e from s1_i (in rowstat) and s2_j.  rowstat (n x heads float4) stays in L2 / Infinity Cache.
template <int TPR, int MODE>
__global__ __launch_bounds__(kThreads) void gat_weights_t_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ s2, int64_t lds2, const float4 *__restrict__ rowstat, int32_t heads,
    float slope, float *__restrict__ alpha_t, int64_t nnz) {
    int64_t j;
    int lane;
    if (!pick_row<TPR>(rows, nlist, j, lane)) return;
    const int k = blockIdx.y;
    const int64_t b = rowptr[j], e = rowptr[j + 1];
    const float a2 = s2[j * lds2 + k];
    float *ak = alpha_t + (int64_t)k * nnz;
    for (int64_t p = b + lane; p < e; p += TPR) {
        const float4 st = rowstat[(int64_t)col[p] * heads + k];
        float r = st.x + a2;
        if (MODE == 0) r = r > 0.f ? r : r * slope;
        ak[p] = (expf(r - st.y) - st.w) * st.z;
    }
}

// The same with ALL KH = heads heads of an entry per pass (KH = 1, 2, 4): one column load and ONE contiguous
// 16 KH-byte gather of the row statistics per entry instead of one of each per head (r03: 3.1 -> see DESIGN 7).
template <int TPR, int MODE, int KH>
__global__ __launch_bounds__(kThreads) void gat_weights_t_heads_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ s2, int64_t lds2, const float4 *__restrict__ rowstat, float slope,
    float *__restrict__ alpha_t, int64_t nnz) {
    int64_t j;
    int lane;
    if (!pick_row<TPR>(rows, nlist, j, lane)) return;
    const int64_t b = rowptr[j], e = rowptr[j + 1];
    float a2[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) a2[k] = s2[j * lds2 + k];
    for (int64_t p = b + lane; p < e; p += TPR) {
        const float4 *st = rowstat + (int64_t)col[p] * KH;
        float4 q[KH];
#pragma unroll
        for (int k = 0; k < KH; ++k) q[k] = st[k];
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            float r = q[k].x + a2[k];
            if (MODE == 0) r = r > 0.f ? r : r * slope;
            alpha_t[(int64_t)k * nnz + p] = (expf(r - q[k].y) - q[k].w) * q[k].z;
        }
    }
}

// KH heads of a row per pass (KH = 1, 2, 4; s2 compact, lds2 % KH == 0): one col load and one
// 4*KH-byte s2 gather per entry serve KH heads, two entries per lane and iteration are in flight.
template <int KH> struct HeadVec;
template <> struct HeadVec<1> { using T = float; };
template <> struct HeadVec<2> { using T = float2; };
template <> struct HeadVec<4> { using T = float4; };

template <int TPR, int MODE, int KH>
__global__ __launch_bounds__(kThreads) void gat_softmax_heads_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ s1, int64_t lds1, const float *__restrict__ s2, int64_t lds2,
    int32_t heads, float slope, float nglobal, float *__restrict__ alpha, float *__restrict__ beta,
    float4 *__restrict__ rowstat, int64_t nnz) {
    using HV = typename HeadVec<KH>::T;
    __shared__ float red[4];
    int64_t i;
    int lane;
    if (!pick_row<TPR>(rows, nlist, i, lane)) return;
    const int kb = blockIdx.y * KH;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    float a[KH], m[KH], sum[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        a[k] = s1[i * lds1 + kb + k];
        m[k] = MODE == 1 ? 0.f : -INFINITY;
        sum[k] = 0.f;
    }
    auto scores = [&](int64_t p, float *r) {
        const HV v = *reinterpret_cast<const HV *>(s2 + (int64_t)col[p] * lds2 + kb);
        const float *vf = reinterpret_cast<const float *>(&v);
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            const float x = a[k] + vf[k];
            r[k] = MODE == 0 ? (x > 0.f ? x : x * slope) : x;
        }
    };
    if (!alpha) {
        // statistics only (r05): ONE pass over the row -- every lane keeps a running maximum and the sum of exponentials relative
        // to it (rescaled when the maximum moves), the lanes' pairs are merged at the end: half the column reads and s2 gathers of
        // the max-then-sum passes below, the same maximum, the sum to fp32 rounding.  Fixed order: deterministic.
        auto push = [&](float &mk, float &sk, float r) {
            const float hi = fmaxf(mk, r), e1 = expf(fminf(mk, r) - hi);       // exp(-|r - m|); exp(-inf) = 0 on the first entry
            sk = r > mk ? sk * e1 + 1.f : sk + e1;
            mk = hi;
        };
        for (int64_t p = b + lane; p < e; p += 2 * TPR) {
            float r0[KH], r1[KH];
            const bool two = p + TPR < e;
            scores(p, r0);
            scores(two ? p + TPR : p, r1);
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                push(m[k], sum[k], r0[k]);
                if (two) push(m[k], sum[k], r1[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            const float ml = m[k];
            m[k] = group_reduce<TPR, true>(ml, red);
            sum[k] = (ml == -INFINITY || sum[k] == 0.f) ? 0.f : sum[k] * expf(ml - m[k]);   // a lane without entries adds nothing
        }
    } else {
    for (int64_t p = b + lane; p < e; p += 2 * TPR) {
        float r0[KH], r1[KH];
        const bool two = p + TPR < e;
        scores(p, r0);
        scores(two ? p + TPR : p, r1);
#pragma unroll
        for (int k = 0; k < KH; ++k) m[k] = fmaxf(m[k], fmaxf(r0[k], r1[k]));
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) m[k] = group_reduce<TPR, true>(m[k], red);
    for (int64_t p = b + lane; p < e; p += 2 * TPR) {
        float r0[KH], r1[KH];
        const bool two = p + TPR < e;
        scores(p, r0);
        scores(two ? p + TPR : p, r1);
#pragma unroll
        for (int k = 0; k < KH; ++k) sum[k] += expf(r0[k] - m[k]) + (two ? expf(r1[k] - m[k]) : 0.f);
    }
    }
    float em[KH], inv[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        sum[k] = group_reduce<TPR, false>(sum[k], red);
        em[k] = 0.f;
        float D = sum[k];
        if (MODE == 1) {
            em[k] = expf(-m[k]);
            D = (nglobal - (float)(e - b)) * em[k] + sum[k];
        }
        inv[k] = D > 0.f ? 1.f / D : 0.f;
    }
    for (int64_t p = b + lane; alpha && p < e; p += 2 * TPR) {   // (alpha == NULL: statistics only, the products recompute it)
        float r0[KH], r1[KH];
        const bool two = p + TPR < e;
        scores(p, r0);
        scores(two ? p + TPR : p, r1);
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            float *ak = alpha + (int64_t)(kb + k) * nnz;
            ak[p] = (expf(r0[k] - m[k]) - em[k]) * inv[k];
            if (two) ak[p + TPR] = (expf(r1[k] - m[k]) - em[k]) * inv[k];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            if (MODE == 1) beta[i * heads + kb + k] = em[k] * inv[k];
            if (rowstat) rowstat[i * heads + kb + k] = make_float4(a[k], m[k], inv[k], em[k]);
        }
    }
}

// (r06: the statistics of the rows above the long-row threshold were also built as one workgroup per CHUNK of a row + a merge of the
//  (maximum, sum) pairs -- pgcn_gat_edge_stats_chunked_f32, commit 19d1915 -- and measured on the Reddit shape: 52.90 / 53.07 ms per epoch
//  at chunks of 4 096 against 52.93 / 52.96 with one workgroup per row, 53.6 at 2 048, 54.9 at 8 192: removed.  profiles/r06_gat_stats.txt)

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
    float *dk = de + k;                    // de is ENTRY-major: de[p * heads + k]
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
                dk[pa * heads] = ga;
            }
            if (vb) {
                gb = (ak[pb] + bi) * (db - ti);
                if (MODE == 0) gb *= (a + s2[cb * lds2 + k]) > 0.f ? 1.f : slope;
                dk[pb * heads] = gb;
            }
            acc += ga + gb;
        }
    }
    acc = group_reduce<TPR, false>(acc, red);
    if (lane == 0) ds1[i * heads + k] = acc;
}

// All heads of a row in one pass: a team of lpe = pow2(F/4) lanes holds one whole row of Z as
// float4 (F = heads*d <= 256); the dot products are reduced inside the d/4-lane segment of each
// head (xor shuffles leave the sum in every lane of the segment).  U entries are in flight per team
// and iteration -- the kernel is bound by gather latency, not by issue -- and lane u of head k's
// segment finishes entry u of the batch, so the alpha loads and the de stores of a head are U
// consecutive addresses.  One gather of the full row per entry instead of one per entry and head.
//
// SLICED (TPR = 64): the structure is stored XCD-sliced (a row's entries grouped by col % 8) and
// slice_off[i][0..8] are the offsets of row i's slices.  Workgroup b runs on XCD b % 8 and works on
// slice b % 8 only, so every XCD's L2 holds one eighth of the Z panel instead of thrashing over all
// of it (the same device the SpMM gather kernel uses); ds1 then holds one partial per (row, slice).
template <int TPR, int MODE, int U, bool SLICED>
__global__ __launch_bounds__(kThreads) void gat_edge_grad_heads_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ s1, int64_t lds1, const float *__restrict__ s2, int64_t lds2,
    const float *__restrict__ alpha, const float *__restrict__ beta, const float *__restrict__ Z, int64_t ldz,
    const float *__restrict__ dOut, int64_t ldo, const float *__restrict__ t, int32_t heads, int32_t d, int32_t lpe,
    float slope, float *__restrict__ de, float *__restrict__ ds1, int64_t nnz, const int32_t *__restrict__ slice_off) {
    __shared__ float red[4];
    int64_t i;
    int lane;
    int64_t b, e;
    int64_t orow;                          // row of ds1 this wave writes
    if constexpr (SLICED) {
        const int sl = blockIdx.x & 7;
        const int64_t li = (int64_t)(blockIdx.x >> 3) * (kThreads / 64) + (threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        if (li >= nlist) return;
        i = rows ? (int64_t)rows[li] : li;
        const int64_t base = rowptr[i];
        b = base + slice_off[i * 9 + sl];
        e = base + slice_off[i * 9 + sl + 1];
        orow = i * 8 + sl;
    } else {
        if (!pick_row<TPR>(rows, nlist, i, lane)) return;
        b = rowptr[i];
        e = rowptr[i + 1];
        orow = i;
    }
    const int nvec = heads * d / 4;
    const int hl = d / 4;                  // lanes per head (>= U)
    const int sub = lane & (lpe - 1);
    const int team = lane / lpe;
    const int nteam = TPR / lpe;
    const int hk = sub / hl;
    const int u_mine = sub % hl;           // which entry of the batch this lane finishes
    const bool live = sub < nvec;
    const bool fin = live && u_mine < U;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) g0 = reinterpret_cast<const float4 *>(dOut + i * ldo)[sub];
    float a = 0.f, ti = 0.f, bi = 0.f;
    if (fin) {
        a = s1[i * lds1 + hk];
        ti = t[i * heads + hk];
        if (MODE == 1) bi = beta[i * heads + hk];
    }
    const float *ak = alpha + (int64_t)hk * nnz;
    float *dk = de + hk;                   // de is ENTRY-major: de[p * heads + k] (the heads of an entry are 4*heads bytes)
    float acc = 0.f;
    // The column ids of the NEXT batch are requested before the row gathers of this one go out, and a lane's alpha /
    // s2 words together with them: the kernel is a latency chain per batch (col -> 1 KB rows -> alpha, s2), r03 took
    // the first and the last link out of it.
    int32_t c[U];
    {
        const int64_t p0 = b + (int64_t)team * U;
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = (e > b) ? col[p0 + u < e ? p0 + u : e - 1] : 0;
    }
    for (int64_t p0 = b + (int64_t)team * U; p0 < e; p0 += (int64_t)nteam * U) {
        const int64_t pn = p0 + (int64_t)nteam * U;
        int32_t cn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cn[u] = col[pn + u < e ? pn + u : e - 1];
        int32_t cm = c[0];
#pragma unroll
        for (int u = 1; u < U; ++u) cm = u_mine == u ? c[u] : cm;
        const int64_t p = p0 + u_mine;
        const bool mine = fin && p < e;
        float al = 0.f, s2v = 0.f;
        if (mine) {
            al = ak[p];
            if (MODE == 0) s2v = s2[(int64_t)cm * lds2 + hk];
        }
        float dot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dot[u] = 0.f;
            if (live) dot[u] = Vec<4>::dot(g0, reinterpret_cast<const float4 *>(Z + (int64_t)c[u] * ldz)[sub]);
        }
        for (int o = hl >> 1; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) dot[u] += __shfl_xor(dot[u], o, 64);
        }
        float dm = dot[0];
#pragma unroll
        for (int u = 1; u < U; ++u) dm = u_mine == u ? dot[u] : dm;
        if (mine) {
            float g = (al + bi) * (dm - ti);
            if (MODE == 0) g *= (a + s2v) > 0.f ? 1.f : slope;
            dk[p * heads] = g;
            acc += g;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = cn[u];
    }
    for (int k = 0; k < heads; ++k) {
        const float v = group_reduce<TPR, false>((fin && hk == k) ? acc : 0.f, red);
        if (lane == 0) ds1[orow * heads + k] = v;
    }
}

// The same pass over the TASKS of the SpMM plan (pgcn_spmm_plan_host: a task is a row, or an XCD slice / a chunk
// of <= `chunk` entries of a long row; workgroup b takes tasks of slice b % nslices = its XCD; longest first).
// One wave per (row, slice) leaves the hub rows of a power-law graph as a tail of a few very long waves
// (Reddit shape: 13 k entries in one wave); chunked tasks balance like the SpMM.  A task's row is found by
// binary search of its first entry in rowptr; its sum over de goes to ds1 directly (the row's only task) or to a
// slot that pgcn_spmm_fixup_f32 adds in slot order (deterministic).
struct GatSeg { int64_t v[PGCN_MAX_SLICES + 1]; };

template <int MODE, int U>
__global__ __launch_bounds__(kThreads) void gat_edge_grad_tasks_kernel(
    const int64_t *__restrict__ rowptr, int64_t nrows, const int32_t *__restrict__ col, const int4 *__restrict__ tasks,
    int64_t ntasks, int32_t nslices, GatSeg seg, const float *__restrict__ s1, int64_t lds1, const float *__restrict__ s2,
    int64_t lds2, const float *__restrict__ alpha, const float *__restrict__ beta, const float *__restrict__ Z, int64_t ldz,
    const float *__restrict__ dOut, int64_t ldo, const float *__restrict__ t, int32_t heads, int32_t d, int32_t lpe,
    float slope, float *__restrict__ de, float *__restrict__ ds1, float *__restrict__ partial, int64_t nnz) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int64_t tid;
    if (nslices > 1) {
        const int sl = blockIdx.x % nslices;
        tid = seg.v[sl] + (int64_t)(blockIdx.x / nslices) * (kThreads / 64) + wave;
        ntasks = seg.v[sl + 1];
    } else {
        tid = (int64_t)blockIdx.x * (kThreads / 64) + wave;
    }
    if (tid >= ntasks) return;
    int64_t b, i;
    int32_t len, dst;
    if (tasks) {
        const int4 tk = tasks[tid];
        b = (int64_t)(((uint64_t)(uint32_t)tk.y << 32) | (uint32_t)tk.x);
        len = tk.z;
        dst = tk.w;
        if (dst < 0) {
            i = ~dst;
        } else {                             // largest i with rowptr[i] <= b (a split row is never empty)
            int64_t lo = 0, hi = nrows - 1;
            while (lo < hi) {
                const int64_t mid = (lo + hi + 1) >> 1;
                if (rowptr[mid] <= b) lo = mid; else hi = mid - 1;
            }
            i = lo;
        }
    } else {
        i = tid;
        b = rowptr[i];
        len = (int32_t)(rowptr[i + 1] - b);
        dst = -1;
    }
    const int64_t e = b + len;
    const int nvec = heads * d / 4;
    const int hl = d / 4;
    const int sub = lane & (lpe - 1);
    const int team = lane / lpe;
    const int nteam = 64 / lpe;
    const int hk = sub / hl;
    const int u_mine = sub % hl;
    const bool live = sub < nvec;
    const bool fin = live && u_mine < U;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) g0 = reinterpret_cast<const float4 *>(dOut + i * ldo)[sub];
    float a = 0.f, ti = 0.f, bi = 0.f;
    if (fin) {
        a = s1[i * lds1 + hk];
        ti = t[i * heads + hk];
        if (MODE == 1) bi = beta[i * heads + hk];
    }
    const float *ak = alpha + (int64_t)hk * nnz;
    float *dk = de + hk;                   // de is ENTRY-major: de[p * heads + k] (the heads of an entry are 4*heads bytes)
    float acc = 0.f;
    // The column ids of the NEXT batch are requested before the row gathers of this one go out, and a lane's alpha /
    // s2 words together with them: the kernel is a latency chain per batch (col -> 1 KB rows -> alpha, s2), r03 took
    // the first and the last link out of it.
    int32_t c[U];
    {
        const int64_t p0 = b + (int64_t)team * U;
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = (e > b) ? col[p0 + u < e ? p0 + u : e - 1] : 0;
    }
    for (int64_t p0 = b + (int64_t)team * U; p0 < e; p0 += (int64_t)nteam * U) {
        const int64_t pn = p0 + (int64_t)nteam * U;
        int32_t cn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cn[u] = col[pn + u < e ? pn + u : e - 1];
        int32_t cm = c[0];
#pragma unroll
        for (int u = 1; u < U; ++u) cm = u_mine == u ? c[u] : cm;
        const int64_t p = p0 + u_mine;
        const bool mine = fin && p < e;
        float al = 0.f, s2v = 0.f;
        if (mine) {
            al = ak[p];
            if (MODE == 0) s2v = s2[(int64_t)cm * lds2 + hk];
        }
        float dot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dot[u] = 0.f;
            if (live) dot[u] = Vec<4>::dot(g0, reinterpret_cast<const float4 *>(Z + (int64_t)c[u] * ldz)[sub]);
        }
        for (int o = hl >> 1; o > 0; o >>= 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) dot[u] += __shfl_xor(dot[u], o, 64);
        }
        float dm = dot[0];
#pragma unroll
        for (int u = 1; u < U; ++u) dm = u_mine == u ? dot[u] : dm;
        if (mine) {
            float g = (al + bi) * (dm - ti);
            if (MODE == 0) g *= (a + s2v) > 0.f ? 1.f : slope;
            dk[p * heads] = g;
            acc += g;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = cn[u];
    }
    float *outp = dst >= 0 ? partial + (int64_t)dst * heads : ds1 + i * heads;
    for (int k = 0; k < heads; ++k) {
        const float v = wave_sum((fin && hk == k) ? acc : 0.f, 64);
        if (lane == 0) outp[k] = v;
    }
}

// out[i, k] = sum over the entries p of row i of src[idx(p) * planes + k], idx(p) = perm[p] or p: `src` is ENTRY-major
// ([nnz][planes]: the planes of an entry are adjacent, so a permuted entry costs ONE gather of 4 * planes bytes -- with
// plane-major storage the gather of ds2 = column sums of de was `planes` random 4-byte gathers per entry, 3.9 of the
// 25 ms of a GAT layer's backward, r02).  KH = planes in {1, 2, 4}: vector loads, all planes per pass; KH = 0: any
// number of planes, one per blockIdx.y.
template <int TPR, int KH>
__global__ __launch_bounds__(kThreads) void csr_row_sums_kernel(
    const int64_t *__restrict__ rowptr, const int64_t *__restrict__ perm, const int32_t *__restrict__ rows,
    int64_t nlist, const float *__restrict__ src, int32_t planes, float *__restrict__ out, int64_t ldo) {
    __shared__ float red[4];
    int64_t i;
    int lane;
    if (!pick_row<TPR>(rows, nlist, i, lane)) return;
    const int64_t b = rowptr[i], e = rowptr[i + 1];
    if constexpr (KH > 0) {
        using HV = typename HeadVec<KH>::T;
        const HV *sv = reinterpret_cast<const HV *>(src);
        float acc[KH];
#pragma unroll
        for (int k = 0; k < KH; ++k) acc[k] = 0.f;
        for (int64_t p = b + lane; p < e; p += TPR) {
            const HV v = sv[perm ? perm[p] : p];
            const float *vf = reinterpret_cast<const float *>(&v);
#pragma unroll
            for (int k = 0; k < KH; ++k) acc[k] += vf[k];
        }
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            const float r = group_reduce<TPR, false>(acc[k], red);
            if (lane == 0) out[i * ldo + k] = r;
        }
    } else {
        const int k = blockIdx.y;
        float acc = 0.f;
        for (int64_t p = b + lane; p < e; p += TPR) acc += src[(perm ? perm[p] : p) * planes + k];
        acc = group_reduce<TPR, false>(acc, red);
        if (lane == 0) out[i * ldo + k] = acc;
    }
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
                                         float *alpha, float *beta, float *rowstat, pgcn_stream_t stream) {
    const char *who = "pgcn_gat_edge_softmax_f32";
    if (nrows < 0 || nnz < 0 || heads < 1 || heads > 65535 || lds1 < heads || lds2 < heads || (mode != 0 && mode != 1))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    const RowLists l{rows_wave, nrows_wave, rows_block, nrows_block};
    int rc = check_lists(who, nrows, l);
    if (rc != PGCN_OK) return rc;
    if (nrows == 0 || l.nwave + l.nblock == 0) return PGCN_OK;
    if (!rowptr || !s1 || (nnz && (!col || !s2 || (!alpha && !rowstat))) || (mode == 1 && !beta))
        return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    if ((uintptr_t)rowstat % 16) return pgcn_set_error2(PGCN_EINVAL, who, "rowstat must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float ng = (float)n_global;
    int kh = 1;
    if (heads % 4 == 0 && lds2 % 4 == 0 && (uintptr_t)s2 % 16 == 0) kh = 4;
    else if (heads % 2 == 0 && lds2 % 2 == 0 && (uintptr_t)s2 % 8 == 0) kh = 2;
    float4 *rs = reinterpret_cast<float4 *>(rowstat);
#define PGCN_SOFTMAX(TPR, MODE, KH, GRID, ROWS, N)                                                                     \
    hipLaunchKernelGGL((gat_softmax_heads_kernel<TPR, MODE, KH>), GRID, dim3(kThreads), 0, s, rowptr, col, ROWS, N, s1, \
                       lds1, s2, lds2, heads, slope, ng, alpha, beta, rs, nnz)
#define PGCN_SOFTMAX_K(TPR, MODE, GRIDFN, ROWS, N)                                      \
    do {                                                                                \
        if (kh == 4) PGCN_SOFTMAX(TPR, MODE, 4, GRIDFN(N, heads / 4), ROWS, N);         \
        else if (kh == 2) PGCN_SOFTMAX(TPR, MODE, 2, GRIDFN(N, heads / 2), ROWS, N);    \
        else PGCN_SOFTMAX(TPR, MODE, 1, GRIDFN(N, heads), ROWS, N);                     \
    } while (0)
    if (l.nwave) {
        if (mode == 0) PGCN_SOFTMAX_K(64, 0, wave_grid, l.wave, l.nwave);
        else PGCN_SOFTMAX_K(64, 1, wave_grid, l.wave, l.nwave);
    }
    if (l.nblock) {
        if (mode == 0) PGCN_SOFTMAX_K(256, 0, block_grid, l.block, l.nblock);
        else PGCN_SOFTMAX_K(256, 1, block_grid, l.block, l.nblock);
    }
#undef PGCN_SOFTMAX_K
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
    const int F4 = heads * d / 4, hl = d / 4;
    if (v4 && F4 <= 64 && (hl & (hl - 1)) == 0) {       // all heads of a row in one pass
        int team = 1;
        while (team < F4) team *= 2;
#define PGCN_EGRAD_H(TPR, MODE, UU, GRID, ROWS, N)                                                                      \
    hipLaunchKernelGGL((gat_edge_grad_heads_kernel<TPR, MODE, UU, false>), GRID, dim3(kThreads), 0, s, rowptr, col, ROWS, N, s1, \
                       lds1, s2, lds2, alpha, beta, Z, ldz, dOut, ldo, t, heads, d, team, slope, de, ds1, nnz, nullptr)
#define PGCN_EGRAD_HU(TPR, GRID, ROWS, N)                                                   \
    do {                                                                                    \
        if (hl >= 8) {                                                                      \
            if (mode == 0) PGCN_EGRAD_H(TPR, 0, 8, GRID, ROWS, N);                          \
            else PGCN_EGRAD_H(TPR, 1, 8, GRID, ROWS, N);                                    \
        } else if (hl >= 4) {                                                               \
            if (mode == 0) PGCN_EGRAD_H(TPR, 0, 4, GRID, ROWS, N);                          \
            else PGCN_EGRAD_H(TPR, 1, 4, GRID, ROWS, N);                                    \
        } else {                                                                            \
            if (mode == 0) PGCN_EGRAD_H(TPR, 0, 1, GRID, ROWS, N);                          \
            else PGCN_EGRAD_H(TPR, 1, 1, GRID, ROWS, N);                                    \
        }                                                                                   \
    } while (0)
        if (l.nwave) PGCN_EGRAD_HU(64, wave_grid(l.nwave, 1), l.wave, l.nwave);
        if (l.nblock) PGCN_EGRAD_HU(256, block_grid(l.nblock, 1), l.block, l.nblock);
#undef PGCN_EGRAD_HU
#undef PGCN_EGRAD_H
    } else {
        if (l.nwave) PGCN_EGRAD_VM(64, wave_grid(l.nwave, heads), l.wave, l.nwave);
        if (l.nblock) PGCN_EGRAD_VM(256, block_grid(l.nblock, heads), l.block, l.nblock);
    }
#undef PGCN_EGRAD_VM
#undef PGCN_EGRAD
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

extern "C" int pgcn_gat_edge_grad_sliced_f32(const int64_t *rowptr, const int32_t *col, const int32_t *slice_off,
                                             int64_t nrows, int64_t nnz, const int32_t *rows, int64_t nlist,
                                             const float *s1, int64_t lds1, const float *s2, int64_t lds2,
                                             const float *alpha, const float *beta, const float *Z, int64_t ldz,
                                             const float *dOut, int64_t ldo, const float *t, int32_t heads, int32_t d,
                                             float slope, int32_t mode, float *de, float *ds1_slices,
                                             pgcn_stream_t stream) {
    const char *who = "pgcn_gat_edge_grad_sliced_f32";
    if (nrows < 0 || nnz < 0 || nlist < 0 || nlist > nrows || heads < 1 || d < 1 || lds1 < heads || lds2 < heads ||
        ldz < (int64_t)heads * d || ldo < (int64_t)heads * d || (mode != 0 && mode != 1))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    const int F4 = heads * d / 4, hl = d / 4;
    const bool v4 = d % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && (uintptr_t)Z % 16 == 0 && (uintptr_t)dOut % 16 == 0;
    if (!v4 || F4 > 64 || (hl & (hl - 1)) != 0)
        return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "needs heads*d <= 256, d a power of two >= 4, 16-byte aligned panels");
    if (nrows == 0 || nlist == 0) return PGCN_OK;
    if (!rowptr || !slice_off || !s1 || !t || !ds1_slices || !dOut || (nnz && (!col || !s2 || !alpha || !Z || !de)) ||
        (mode == 1 && !beta))
        return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    const int64_t blocks = (nlist + 3) / 4 * 8;
    if (blocks > 0x7fffffffLL) return pgcn_set_error2(PGCN_EINVAL, who, "too many rows");
    hipStream_t s = (hipStream_t)stream;
    int team = 1;
    while (team < F4) team *= 2;
    const dim3 grid((unsigned)blocks);
#define PGCN_EGRAD_S(MODE, UU)                                                                                        \
    hipLaunchKernelGGL((gat_edge_grad_heads_kernel<64, MODE, UU, true>), grid, dim3(kThreads), 0, s, rowptr, col, rows, \
                       nlist, s1, lds1, s2, lds2, alpha, beta, Z, ldz, dOut, ldo, t, heads, d, team, slope, de,       \
                       ds1_slices, nnz, slice_off)
    if (hl >= 8) {
        if (mode == 0) PGCN_EGRAD_S(0, 8);
        else PGCN_EGRAD_S(1, 8);
    } else if (hl >= 4) {
        if (mode == 0) PGCN_EGRAD_S(0, 4);
        else PGCN_EGRAD_S(1, 4);
    } else {
        if (mode == 0) PGCN_EGRAD_S(0, 1);
        else PGCN_EGRAD_S(1, 1);
    }
#undef PGCN_EGRAD_S
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

extern "C" int pgcn_gat_edge_grad_tasks_f32(const int64_t *rowptr, const int32_t *col, int64_t nrows, int64_t nnz,
                                            const int32_t *tasks, int64_t ntasks, const int64_t *seg, int32_t nslices,
                                            const int32_t *fix, int64_t nfix, const float *s1, int64_t lds1,
                                            const float *s2, int64_t lds2, const float *alpha, const float *beta,
                                            const float *Z, int64_t ldz, const float *dOut, int64_t ldo, const float *t,
                                            int32_t heads, int32_t d, float slope, int32_t mode, float *de, float *ds1,
                                            float *partial_ws, int64_t partial_ws_elems, int64_t nslots,
                                            pgcn_stream_t stream) {
    const char *who = "pgcn_gat_edge_grad_tasks_f32";
    if (nrows < 0 || nnz < 0 || ntasks < 0 || nfix < 0 || nslots < 0 || heads < 1 || d < 1 || lds1 < heads || lds2 < heads ||
        ldz < (int64_t)heads * d || ldo < (int64_t)heads * d || (mode != 0 && mode != 1) || nslices < 1 ||
        nslices > PGCN_MAX_SLICES || (nslices > 1 && (!seg || !tasks)))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    const int F4 = heads * d / 4, hl = d / 4;
    const bool v4 = d % 4 == 0 && ldz % 4 == 0 && ldo % 4 == 0 && (uintptr_t)Z % 16 == 0 && (uintptr_t)dOut % 16 == 0;
    if (!v4 || F4 > 64 || (hl & (hl - 1)) != 0)
        return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "needs heads*d <= 256, d a power of two >= 4, 16-byte aligned panels");
    const int64_t nt = tasks ? ntasks : nrows;
    if (nrows == 0 || nt == 0) return PGCN_OK;
    if (!rowptr || !s1 || !t || !ds1 || !dOut || (nnz && (!col || !s2 || !alpha || !Z || !de)) || (mode == 1 && !beta) ||
        (nfix > 0 && !fix) || (nslots > 0 && !partial_ws))
        return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    if (partial_ws_elems < nslots * (int64_t)heads) return pgcn_set_error2(PGCN_ENOMEM, who, "partial work-space too small");
    GatSeg sg{};
    int64_t grid;
    if (nslices > 1) {
        if (seg[0] != 0 || seg[nslices] != ntasks) return pgcn_set_error2(PGCN_EINVAL, who, "seg does not cover the task list");
        int64_t longest = 0;
        for (int q = 0; q <= nslices; ++q) sg.v[q] = seg[q];
        for (int q = 0; q < nslices; ++q) longest = sg.v[q + 1] - sg.v[q] > longest ? sg.v[q + 1] - sg.v[q] : longest;
        grid = ((longest + 3) / 4) * nslices;
    } else {
        grid = (nt + 3) / 4;
    }
    if (grid > 0x7fffffffLL) return pgcn_set_error2(PGCN_EINVAL, who, "too many tasks");
    hipStream_t s = (hipStream_t)stream;
    int team = 1;
    while (team < F4) team *= 2;
    const int4 *t4 = reinterpret_cast<const int4 *>(tasks);
#define PGCN_EGRAD_T(MODE, UU)                                                                                       \
    hipLaunchKernelGGL((gat_edge_grad_tasks_kernel<MODE, UU>), dim3((unsigned)grid), dim3(kThreads), 0, s, rowptr, nrows, \
                       col, t4, nt, nslices, sg, s1, lds1, s2, lds2, alpha, beta, Z, ldz, dOut, ldo, t, heads, d, team,   \
                       slope, de, ds1, partial_ws, nnz)
    if (hl >= 8) {
        if (mode == 0) PGCN_EGRAD_T(0, 8);
        else PGCN_EGRAD_T(1, 8);
    } else if (hl >= 4) {
        if (mode == 0) PGCN_EGRAD_T(0, 4);
        else PGCN_EGRAD_T(1, 4);
    } else {
        if (mode == 0) PGCN_EGRAD_T(0, 1);
        else PGCN_EGRAD_T(1, 1);
    }
#undef PGCN_EGRAD_T
    PGCN_HIP_CHECK(hipGetLastError());
    if (nfix > 0) return pgcn_spmm_fixup_f32(fix, nfix, nullptr, nullptr, partial_ws, ds1, heads, heads, 0, stream);
    return PGCN_OK;
}

extern "C" int pgcn_gat_edge_weights_t_f32(const int64_t *rowptr_t, const int32_t *col_t, int64_t nrows_t, int64_t nnz,
                                           const int32_t *rows_wave, int64_t nrows_wave, const int32_t *rows_block,
                                           int64_t nrows_block, const float *s2, int64_t lds2, const float *rowstat,
                                           int32_t heads, float slope, int32_t mode, float *alpha_t,
                                           pgcn_stream_t stream) {
    const char *who = "pgcn_gat_edge_weights_t_f32";
    if (nrows_t < 0 || nnz < 0 || heads < 1 || heads > 65535 || lds2 < heads || (mode != 0 && mode != 1))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    const RowLists l{rows_wave, nrows_wave, rows_block, nrows_block};
    int rc = check_lists(who, nrows_t, l);
    if (rc != PGCN_OK) return rc;
    if (nrows_t == 0 || nnz == 0 || l.nwave + l.nblock == 0) return PGCN_OK;
    if (!rowptr_t || !col_t || !s2 || !rowstat || !alpha_t) return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    if ((uintptr_t)rowstat % 16) return pgcn_set_error2(PGCN_EINVAL, who, "rowstat must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float4 *rs = reinterpret_cast<const float4 *>(rowstat);
#define PGCN_WT(TPR, MODE, GRID, ROWS, N)                                                                         \
    hipLaunchKernelGGL((gat_weights_t_kernel<TPR, MODE>), GRID, dim3(kThreads), 0, s, rowptr_t, col_t, ROWS, N, s2, \
                       lds2, rs, heads, slope, alpha_t, nnz)
#define PGCN_WTH(TPR, MODE, KH, GRID, ROWS, N)                                                                       \
    hipLaunchKernelGGL((gat_weights_t_heads_kernel<TPR, MODE, KH>), GRID, dim3(kThreads), 0, s, rowptr_t, col_t, ROWS, \
                       N, s2, lds2, rs, slope, alpha_t, nnz)
#define PGCN_WTK(TPR, GRID1, ROWS, N)                                                                  \
    do {                                                                                               \
        if (heads == 4) { if (mode == 0) PGCN_WTH(TPR, 0, 4, GRID1, ROWS, N); else PGCN_WTH(TPR, 1, 4, GRID1, ROWS, N); } \
        else if (heads == 2) { if (mode == 0) PGCN_WTH(TPR, 0, 2, GRID1, ROWS, N); else PGCN_WTH(TPR, 1, 2, GRID1, ROWS, N); } \
        else { if (mode == 0) PGCN_WTH(TPR, 0, 1, GRID1, ROWS, N); else PGCN_WTH(TPR, 1, 1, GRID1, ROWS, N); } \
    } while (0)
    const bool allheads = heads == 1 || heads == 2 || heads == 4;      // all heads of an entry per pass
    if (l.nwave) {
        if (allheads) PGCN_WTK(64, wave_grid(l.nwave, 1), l.wave, l.nwave);
        else if (mode == 0) PGCN_WT(64, 0, wave_grid(l.nwave, heads), l.wave, l.nwave);
        else PGCN_WT(64, 1, wave_grid(l.nwave, heads), l.wave, l.nwave);
    }
    if (l.nblock) {
        if (allheads) PGCN_WTK(256, block_grid(l.nblock, 1), l.block, l.nblock);
        else if (mode == 0) PGCN_WT(256, 0, block_grid(l.nblock, heads), l.block, l.nblock);
        else PGCN_WT(256, 1, block_grid(l.nblock, heads), l.block, l.nblock);
    }
#undef PGCN_WTK
#undef PGCN_WTH
#undef PGCN_WT
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

// ---- per-row, per-head dot products of the backward (r05) -------------------------------------------------------------------------
// t[i][k] = <dOut_i, out_i> over head k (PGAT.py's softmax backward needs it per row), and -- with the forward pass's second
// accumulator VC = [V | C] -- ds1[i][k] = <dOut_i, V_i>_k - t[i][k] * C[i][k].  One wave per row, a lane holds 4 consecutive
// features, a head = d / 4 consecutive lanes (a power of two): ONE pass over dOut, out and V instead of the four element-wise /
// reduction kernels the tensor expressions launch (2 x 239 MB written and read back as temporaries at the benchmark shape).
namespace {
__global__ __launch_bounds__(256) void gat_row_dots_kernel(const float *__restrict__ dOut, int64_t ldo, const float *__restrict__ out,
                                                           int64_t ldout, const float *__restrict__ VC, int64_t ldv, int64_t n,
                                                           int heads, int d, float *__restrict__ t, float *__restrict__ ds1) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int F = heads * d, hl = d >> 2;             // lanes per head
    const int f0 = lane * 4;
    float a = 0.f, b = 0.f;
    if (f0 < F) {
        const float4 g = *reinterpret_cast<const float4 *>(dOut + row * ldo + f0);
        const float4 o = *reinterpret_cast<const float4 *>(out + row * ldout + f0);
        a = ((g.x * o.x + g.y * o.y) + g.z * o.z) + g.w * o.w;
        if (VC) {
            const float4 v = *reinterpret_cast<const float4 *>(VC + row * ldv + f0);
            b = ((g.x * v.x + g.y * v.y) + g.z * v.z) + g.w * v.w;
        }
    }
    for (int o = hl >> 1; o > 0; o >>= 1) {           // butterfly inside a head's lanes: every lane ends with the head's sum
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (f0 < F && (lane & (hl - 1)) == 0) {
        const int k = lane / hl;
        t[row * heads + k] = a;
        if (VC) ds1[row * heads + k] = b - a * VC[row * ldv + F + k];
    }
}
}  // namespace

extern "C" int pgcn_gat_row_dots_f32(const float *dOut, int64_t ldo, const float *out, int64_t ldout, const float *VC, int64_t ldv,
                                     int64_t n, int32_t heads, int32_t d, float *t, float *ds1, pgcn_stream_t stream) {
    const char *who = "pgcn_gat_row_dots_f32";
    const int hl = d / 4;
    if (n < 0 || heads < 1 || d < 4 || d % 4 || (hl & (hl - 1)) || heads * d > 256 || ldo < heads * d || ldout < heads * d ||
        (VC && ldv < heads * d + heads))
        return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "heads * d <= 256, d / 4 a power of two");
    if (n == 0) return PGCN_OK;
    if (!dOut || !out || !t || (VC && !ds1)) return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    if ((uintptr_t)dOut % 16 || (uintptr_t)out % 16 || ldo % 4 || ldout % 4 || (VC && ((uintptr_t)VC % 16 || ldv % 4)))
        return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "rows must be 16-byte pieces");
    hipLaunchKernelGGL(gat_row_dots_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dOut, ldo, out, ldout, VC,
                       ldv, n, heads, d, t, ds1);
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
    const bool v4 = planes == 4 && (uintptr_t)src % 16 == 0, v2 = planes == 2 && (uintptr_t)src % 8 == 0, v1 = planes == 1;
#define PGCN_RS(TPR, KH, GRID, ROWS, N)                                                                        \
    hipLaunchKernelGGL((csr_row_sums_kernel<TPR, KH>), GRID, dim3(kThreads), 0, s, rowptr, perm, ROWS, N, src, planes, \
                       out, ldo)
    if (l.nwave) {
        if (v4) PGCN_RS(64, 4, wave_grid(l.nwave, 1), l.wave, l.nwave);
        else if (v2) PGCN_RS(64, 2, wave_grid(l.nwave, 1), l.wave, l.nwave);
        else if (v1) PGCN_RS(64, 1, wave_grid(l.nwave, 1), l.wave, l.nwave);
        else PGCN_RS(64, 0, wave_grid(l.nwave, planes), l.wave, l.nwave);
    }
    if (l.nblock) {
        if (v4) PGCN_RS(256, 4, block_grid(l.nblock, 1), l.block, l.nblock);
        else if (v2) PGCN_RS(256, 2, block_grid(l.nblock, 1), l.block, l.nblock);
        else if (v1) PGCN_RS(256, 1, block_grid(l.nblock, 1), l.block, l.nblock);
        else PGCN_RS(256, 0, block_grid(l.nblock, planes), l.block, l.nblock);
    }
#undef PGCN_RS
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
