// pgcn_spmm_heads.hip -- multi-head weighted SpMM for the GAT path (gfx950), fp32.
//
//   C[i, k*d .. (k+1)*d) (+)= sum_e alpha[k][e] * B[col[e], k*d .. (k+1)*d)      for all K heads at once
//
// The reference computes `attention @ Z` with a DENSE n x n attention matrix
// (/root/reference/GPU/PGAT.py:148); r01 ran the sparse equivalent as K launches of the CSR SpMM, one
// per head, each re-reading the structure and gathering a d-wide slice of every neighbour row.  Here
// ONE wave owns a task (a row, or an XCD slice / chunk of a long row: the plan of pgcn_spmm_plan_host)
// and gathers the whole K*d <= 256-wide row as one fully coalesced segment (64 lanes x float4 = 1 KiB);
// a lane's weight is the alpha plane of the head its four features belong to.
//   * col and the K alpha values of 64 entries are fetched lane-parallel (coalesced, non-temporal),
//     parked in LDS (wave private, no barrier) and read back as (col: broadcast, alpha: one word per
//     head group) -- two ds_read_b32 per entry instead of K+1 cross-lane reads;
//   * gathers go out in unpredicated batches of 8 (ragged batches re-read the last referenced row and
//     are zeroed by a select: nothing the task does not reference is ever combined);
//   * split rows leave partial sums in the work-space, combined in slot order by pgcn_spmm_fixup_f32.
//
// RECOMPUTE (r03, the transposed product dZ = A_alpha^T . dOut of the backward pass): the weights are not read from
// planes but recomputed per entry from the softmax's row statistics -- entry (row j, col i) of A^T is entry (i, j) of
// A: alpha = (exp(e - m_i) - em_i) / D_i with e from s1_i (in rowstat[i]) and s2_j -- the arithmetic of
// pgcn_gat_edge_weights_t_f32 moved into this kernel's lane-parallel prefetch stage, bit for bit.  The alpha^T planes
// (4 bytes per entry and head written and read back) and the extra walk over the structure disappear.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgcn_device.h"
#include "pgcn_internal.h"

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kBatch = 8;
constexpr int kMaxHeads = 8;

struct SliceSeg { int64_t v[PGCN_MAX_SLICES + 1]; };

struct Recompute {               // RECOMPUTE: alpha from (rowstat of the entry's column, s2 of the task's row)
    const float4 *rowstat;       // [ncols x KH] (s1, m, 1/D, exp(-m) or 0)
    const float *s2;             // [nrows x lds2]
    int64_t lds2;
    int64_t nrows;
    float slope;
    int32_t mode;                // 0: LeakyReLU(slope) on the raw score, 1: none (reference-literal mode)
};

template <int KH, bool RECOMPUTE>
__global__ __launch_bounds__(kThreads, 4) void spmm_heads_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ alpha,
    int64_t plane, const int4 *__restrict__ tasks, int64_t ntasks, const float *__restrict__ B, int64_t ldb,
    float *__restrict__ C, int64_t ldc, int32_t F, int32_t d, float *__restrict__ partial, uint32_t flags,
    int32_t nslices, SliceSeg seg, Recompute rc) {
    __shared__ float park[kWaves][64 * (KH + 1)];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int64_t tid;
    if (nslices > 1) {   // workgroup b runs on XCD b % 8 and takes tasks of slice b % nslices (see pgcn_spmm.hip)
        const int slice = blockIdx.x % nslices;
        tid = seg.v[slice] + (int64_t)(blockIdx.x / nslices) * kWaves + wave;
        ntasks = seg.v[slice + 1];
    } else {
        tid = (int64_t)blockIdx.x * kWaves + wave;
    }
    const bool tact = tid < ntasks;
    int32_t len = 0, dst = -1;
    int64_t kbeg = 0;
    if (tact) {
        if (tasks) {
            const int4 t = tasks[tid];
            kbeg = (int64_t)(((uint64_t)(uint32_t)t.y << 32) | (uint32_t)t.x);
            len = t.z;
            dst = t.w;
        } else {
            kbeg = rowptr[tid];
            len = (int32_t)(rowptr[tid + 1] - kbeg);
            dst = ~(int32_t)tid;
        }
    }
    const int fcol = lane * 4;
    const bool fact = fcol < F;
    const int head = fact ? fcol / d : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float *mine = park[wave];
    const int last = len > 0 ? len - 1 : 0;
    float a2[KH];                          // RECOMPUTE: s2 of this task's row
#pragma unroll
    for (int k = 0; k < KH; ++k) a2[k] = 0.f;
    if constexpr (RECOMPUTE) {
        if (len > 0) {
            int64_t row;
            if (dst < 0) {
                row = ~dst;
            } else {                         // largest row with rowptr[row] <= kbeg (a split row is never empty)
                int64_t lo = 0, hi = rc.nrows - 1;
                while (lo < hi) {
                    const int64_t mid = (lo + hi + 1) >> 1;
                    if (rowptr[mid] <= kbeg) lo = mid; else hi = mid - 1;
                }
                row = lo;
            }
#pragma unroll
            for (int k = 0; k < KH; ++k) a2[k] = rc.s2[row * rc.lds2 + k];
        }
    }
    // the KH weights of entry `idx` (column c): from the planes, or recomputed from the statistics of row c
    auto weights = [&](int64_t idx, int32_t c, float (&w)[KH]) {
        if constexpr (RECOMPUTE) {
            const float4 *st = rc.rowstat + (int64_t)c * KH;
            float4 q[KH];
#pragma unroll
            for (int k = 0; k < KH; ++k) q[k] = st[k];
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                float r = q[k].x + a2[k];
                if (rc.mode == 0) r = r > 0.f ? r : r * rc.slope;
                w[k] = (expf(r - q[k].y) - q[k].w) * q[k].z;
            }
        } else {
#pragma unroll
            for (int k = 0; k < KH; ++k) w[k] = __builtin_nontemporal_load(alpha + (int64_t)k * plane + idx);
        }
    };
    // prefetch of the first 64 (col, alpha) tuples: unconditional, index clamped into the task
    int32_t nc = 0;
    float na[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) na[k] = 0.f;
    if (len > 0) {
        const int e = lane < last ? lane : last;
        nc = __builtin_nontemporal_load(col + kbeg + e);
        weights(kbeg + e, nc, na);
    }
    for (int base = 0; base < len; base += 64) {   // wave-uniform: one task per wave
        __builtin_amdgcn_wave_barrier();
        mine[lane * (KH + 1)] = __int_as_float(nc);
#pragma unroll
        for (int k = 0; k < KH; ++k) mine[lane * (KH + 1) + 1 + k] = na[k];
        __builtin_amdgcn_wave_barrier();
        if (base + 64 < len) {
            int e = base + 64 + lane;
            e = e < last ? e : last;
            nc = __builtin_nontemporal_load(col + kbeg + e);
            weights(kbeg + e, nc, na);
        }
        const int cnt = min(64, len - base);
        for (int e0 = 0; e0 < cnt; e0 += kBatch) {
            int32_t c[kBatch];
            float w[kBatch];
            float x[kBatch][4];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int e = e0 + u < cnt ? e0 + u : cnt - 1;       // ragged tail: the task's last referenced row
                c[u] = __float_as_int(mine[e * (KH + 1)]);
                w[u] = mine[e * (KH + 1) + 1 + head];
            }
            if (fact) {
#pragma unroll
                for (int u = 0; u < kBatch; ++u) vload<4>(x[u], B + (int64_t)c[u] * ldb + fcol);
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const bool keep = e0 + u < cnt;
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[v] = fmaf(keep ? w[u] : 0.f, keep ? x[u][v] : 0.f, acc[v]);
                }
            }
        }
    }
    if (tact && fact) {
        if (dst >= 0) {
            vstore<4>(partial + (int64_t)dst * F + fcol, acc);
        } else {
            float *cp = C + (int64_t)(~dst) * ldc + fcol;
            if (flags & PGCN_SPMM_ACCUMULATE) {
                float old[4];
                vload<4>(old, cp);
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] += old[v];
            }
            vstore<4>(cp, acc);
        }
    }
}

}  // namespace

namespace {
int launch_heads(const char *who, const int64_t *rowptr, const int32_t *col, const float *alpha, int64_t plane_stride,
                 const Recompute *rc, int32_t heads, int32_t d, int64_t nrows, const int32_t *tasks, int64_t ntasks,
                 const int64_t *seg, int32_t nslices, const int32_t *fix, int64_t nfix, const float *B, int64_t ldb,
                 float *C, int64_t ldc, float *partial_ws, int64_t partial_ws_elems, int64_t nslots, uint32_t flags,
                 pgcn_stream_t stream) {
    const int64_t F = (int64_t)heads * d;
    if (heads <= 0 || d <= 0 || nrows < 0 || ntasks < 0 || nfix < 0 || nslices < 1 || nslices > PGCN_MAX_SLICES ||
        ldb < F || ldc < F || (nslices > 1 && (!seg || !tasks)))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad sizes");
    if (heads > kMaxHeads || F > 256 || d % 4 || ldb % 4 || ldc % 4 || (uintptr_t)B % 16 || (uintptr_t)C % 16 ||
        (uintptr_t)partial_ws % 16)
        return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "needs heads <= 8, heads * d <= 256, d % 4 == 0 and "
                                                       "16-byte aligned operands (use one pgcn_spmm_csr_plan_f32 per head)");
    const int64_t nt = tasks ? ntasks : nrows;
    if (nt == 0) return PGCN_OK;
    if (!rowptr || !col || (!alpha && !rc) || !B || !C || (nfix > 0 && !fix) || (nslots > 0 && !partial_ws))
        return pgcn_set_error2(PGCN_EINVAL, who, "null pointer");
    if (rc && (!rc->rowstat || !rc->s2 || rc->lds2 < heads || (uintptr_t)rc->rowstat % 16 || (rc->mode != 0 && rc->mode != 1)))
        return pgcn_set_error2(PGCN_EINVAL, who, "bad row statistics / s2 / mode");
    if (nslots < 0 || partial_ws_elems < nslots * F)
        return pgcn_set_error2(PGCN_ENOMEM, who, "partial work-space too small");
    SliceSeg sg{};
    int64_t grid;
    if (nslices > 1) {
        if (seg[0] != 0 || seg[nslices] != ntasks)
            return pgcn_set_error2(PGCN_EINVAL, who, "seg does not cover the task list");
        int64_t longest = 0;
        for (int i = 0; i <= nslices; ++i) sg.v[i] = seg[i];
        for (int i = 0; i < nslices; ++i) longest = sg.v[i + 1] - sg.v[i] > longest ? sg.v[i + 1] - sg.v[i] : longest;
        grid = ((longest + kWaves - 1) / kWaves) * nslices;
    } else {
        grid = (nt + kWaves - 1) / kWaves;
    }
    if (grid > 0x7fffffffLL) return pgcn_set_error2(PGCN_EINVAL, who, "too many tasks for one launch");
    hipStream_t s = (hipStream_t)stream;
    const int4 *t4 = reinterpret_cast<const int4 *>(tasks);
    Recompute r0{};
    if (rc) r0 = *rc;
#define PGCN_HEADS(KH)                                                                                              \
    case KH:                                                                                                        \
        if (rc)                                                                                                     \
            hipLaunchKernelGGL((spmm_heads_kernel<KH, true>), dim3((unsigned)grid), dim3(kThreads), 0, s, rowptr, col,  \
                               alpha, plane_stride, t4, nt, B, ldb, C, ldc, (int32_t)F, d, partial_ws, flags, nslices, \
                               sg, r0);                                                                              \
        else                                                                                                        \
            hipLaunchKernelGGL((spmm_heads_kernel<KH, false>), dim3((unsigned)grid), dim3(kThreads), 0, s, rowptr, col, \
                               alpha, plane_stride, t4, nt, B, ldb, C, ldc, (int32_t)F, d, partial_ws, flags, nslices, \
                               sg, r0);                                                                              \
        break;
    switch (heads) {
        PGCN_HEADS(1) PGCN_HEADS(2) PGCN_HEADS(3) PGCN_HEADS(4) PGCN_HEADS(5) PGCN_HEADS(6) PGCN_HEADS(7) PGCN_HEADS(8)
    }
#undef PGCN_HEADS
    PGCN_HIP_CHECK(hipGetLastError());
    if (nfix > 0)
        return pgcn_spmm_fixup_f32(fix, nfix, nullptr, nullptr, partial_ws, C, ldc, (int32_t)F,
                                   flags & PGCN_SPMM_ACCUMULATE, stream);
    return PGCN_OK;
}
}  // namespace

extern "C" int pgcn_spmm_heads_f32(const int64_t *rowptr, const int32_t *col, const float *alpha, int64_t plane_stride,
                                   int32_t heads, int32_t d, int64_t nrows, const int32_t *tasks, int64_t ntasks,
                                   const int64_t *seg, int32_t nslices, const int32_t *fix, int64_t nfix,
                                   const float *B, int64_t ldb, float *C, int64_t ldc, float *partial_ws,
                                   int64_t partial_ws_elems, int64_t nslots, uint32_t flags, pgcn_stream_t stream) {
    return launch_heads("pgcn_spmm_heads_f32", rowptr, col, alpha, plane_stride, nullptr, heads, d, nrows, tasks, ntasks, seg,
                        nslices, fix, nfix, B, ldb, C, ldc, partial_ws, partial_ws_elems, nslots, flags, stream);
}

extern "C" int pgcn_spmm_heads_recompute_f32(const int64_t *rowptr, const int32_t *col, const float *rowstat,
                                             const float *s2, int64_t lds2, float slope, int32_t mode, int32_t heads,
                                             int32_t d, int64_t nrows, const int32_t *tasks, int64_t ntasks,
                                             const int64_t *seg, int32_t nslices, const int32_t *fix, int64_t nfix,
                                             const float *B, int64_t ldb, float *C, int64_t ldc, float *partial_ws,
                                             int64_t partial_ws_elems, int64_t nslots, uint32_t flags,
                                             pgcn_stream_t stream) {
    const Recompute rc{reinterpret_cast<const float4 *>(rowstat), s2, lds2, nrows, slope, mode};
    return launch_heads("pgcn_spmm_heads_recompute_f32", rowptr, col, nullptr, 0, &rc, heads, d, nrows, tasks, ntasks, seg,
                        nslices, fix, nfix, B, ldb, C, ldc, partial_ws, partial_ws_elems, nslots, flags, stream);
}
