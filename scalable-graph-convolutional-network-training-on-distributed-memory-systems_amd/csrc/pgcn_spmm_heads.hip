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
//
// GRAD (r03, with RECOMPUTE): the EDGE GRADIENT in the same pass.  The backward pass needs, for every stored (i, j),
// dp_ij = <dOut_i, Z_j> (pgcn_gat_edge_grad_f32: row i gathers Z_j) and alpha_ij dOut_i summed into dZ_j (this kernel:
// row j gathers dOut_i) -- the same pairs of 1 KB rows, gathered twice.  On the transposed structure Z_j is the task's
// own row (registers) and dOut_i is gathered anyway, so the dot products cost four FMAs and a 16-lane reduction per
// entry: de_ij = (alpha_ij + beta_i)(dp_ij - t_i)[x LeakyReLU'] is written ENTRY-major in the storage order of the
// TRANSPOSED structure and its row sums -- ds2_j -- leave with the row (columns [F, F + heads) of C).  One whole gather
// pass over the graph (the edge-gradient kernel) disappears; ds1_i = the column sums of de are
// pgcn_csr_row_sums_f32 over the forward structure with the inverse permutation.
//
// FWD (r03, with RECOMPUTE): the FORWARD product with recomputed weights and a second accumulator.  alpha_ij comes from
// the row statistics of the task's own row and s2 of the entry's column (no alpha planes: the softmax kernel stops
// after its statistics), and the same gathered Z_j also feeds V_i = sum_j c_ij Z_j and C_i = sum_j c_ij with
// c_ij = alpha_ij LeakyReLU'(s1_i + s2_j) (mode 0) or alpha_ij + beta_i (mode 1).  Since ds1_i = sum_j de_ij =
// <dOut_i, V_i> - t_i C_i, the backward pass needs no per-entry gradient array at all: de is never written, its
// column sums (a pass of 16-byte random gathers through the transpose permutation) disappear.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgcn_device.h"
#include "pgcn_internal.h"

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kBatch = 8;
constexpr int kMaxHeads = 8;

using f32x2 = __attribute__((ext_vector_type(2))) float;

// x of another lane of the same 16-lane row through the DPP network (CTRL: 0x120 + n = row_ror:n, 0x00-0xFF = quad_perm)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// lane ^ 4 inside a 16-lane row: row_shl:4 serves the lanes of banks 0 and 2, row_shr:4 those of banks 1 and 3
__device__ __forceinline__ float dpp_xor4(float x) {
    const int v = __float_as_int(x);
    int r = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
    r = __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);
    return __int_as_float(r);
}

struct SliceSeg { int64_t v[PGCN_MAX_SLICES + 1]; };

struct Recompute {               // RECOMPUTE: alpha from (rowstat of the entry's column, s2 of the task's row)
    const float4 *rowstat;       // [ncols x KH] (s1, m, 1/D, exp(-m) or 0)
    const float *s2;             // [nrows x lds2]
    int64_t lds2;
    int64_t nrows;
    float slope;
    int32_t mode;                // 0: LeakyReLU(slope) on the raw score, 1: none (reference-literal mode)
};

struct EdgeGrad {                // GRAD: what the edge gradient needs on top of Recompute
    const float *Z;              // [nrows x ldz] Z_j of the task's row
    int64_t ldz;
    const float *t;              // [ncols x KH] t_i = <dOut_i, out_i>
    float *de;                   // [nnz x KH] entry-major, storage order of this (transposed) structure
    int32_t pw;                  // width of a partial / output row: F + heads rounded up to 4
};

struct Forward2 {                // FWD: the second output of the forward product
    float *C2;                   // [nrows x ldc2]: V in columns [0, F), C in [F, F + KH), zeros up to pw2
    int64_t ldc2;
    float *partial2;             // slot rows of pw2 floats (behind the F-wide slot rows of the first output)
    int32_t pw2;                 // F + heads rounded up to 4
};

template <int KH, bool RECOMPUTE, bool GRAD, bool FWD>
__global__ __launch_bounds__(kThreads, FWD ? 5 : 4) void spmm_heads_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ alpha,
    int64_t plane, const int4 *__restrict__ tasks, int64_t ntasks, const float *__restrict__ B, int64_t ldb,
    float *__restrict__ C, int64_t ldc, int32_t F, int32_t d, float *__restrict__ partial, uint32_t flags,
    int32_t nslices, SliceSeg seg, Recompute rc, EdgeGrad eg, Forward2 f2) {
    static_assert(!GRAD || RECOMPUTE, "the fused edge gradient recomputes its weights");
    static_assert(!FWD || (RECOMPUTE && !GRAD), "the two-accumulator forward product recomputes its weights");
    // parked words per entry: col, alpha[KH] (GRAD: A[KH], t[KH]; FWD: c[KH])
    constexpr int PS = GRAD ? 3 * KH + 1 : (FWD ? 2 * KH + 1 : KH + 1);
    __shared__ float park[kWaves][64 * PS];
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
    const int fsafe = fact ? fcol : 0;
    const int head = fact ? fcol / d : 0;
    f32x2 accl = {0.f, 0.f}, acch = {0.f, 0.f};        // this lane's four features of the output row
    f32x2 vl = {0.f, 0.f}, vh = {0.f, 0.f};            // FWD: ... and of V
    float csum[KH];                                    // FWD: this lane's share of C_i
#pragma unroll
    for (int k = 0; k < KH; ++k) csum[k] = 0.f;
    float *mine = park[wave];
    const int last = len > 0 ? len - 1 : 0;
    float a2[KH];                          // RECOMPUTE: s2 of this task's row
#pragma unroll
    for (int k = 0; k < KH; ++k) a2[k] = 0.f;
    int64_t row = 0;
    if constexpr (RECOMPUTE) {
        if (len > 0) {
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
            if constexpr (!FWD) {
#pragma unroll
                for (int k = 0; k < KH; ++k) a2[k] = rc.s2[row * rc.lds2 + k];
            }
        }
    }
    // GRAD: this lane's four features of Z_j, the lanes of a head, and which entry of a batch the lane finishes
    f32x2 zl = {0.f, 0.f}, zh = {0.f, 0.f};
    const int hl = d >> 2;                 // lanes per head (the host admits powers of two >= kBatch only)
    // which entry of a batch this lane finishes, and whether it is the one lane that does (hl == 16: see finish())
    const int u_mine = hl == 16 ? (lane & 15) >> 1 : lane & (hl - 1);
    const bool writer = hl == 16 ? !(lane & 1) : u_mine < kBatch;
    float acc2 = 0.f;
    if constexpr (GRAD) {
        if (len > 0 && fact) {
            float zr[4];
            vload<4>(zr, eg.Z + row * eg.ldz + fcol);
            zl = f32x2{zr[0], zr[1]};
            zh = f32x2{zr[2], zr[3]};
        }
    }
    // the KH weights of entry `idx` (column c): from the planes, or recomputed from the statistics of row c;
    // GRAD: also ga = (alpha + beta) [x LeakyReLU'] and gt = t of row c, so that de = ga (dp - gt)
    auto weights = [&](int64_t idx, int32_t c, float (&w)[KH], float (&ga)[KH], float (&gt)[KH]) {
        if constexpr (FWD) {               // row statistics of the task's row, s2 of the entry's column; ga = c_ij
            float sv[KH];
            float4 qrow[KH];               // (one line, the same for all lanes, re-read per 64 entries: no registers
#pragma unroll                             //  held across the gather loop)
            for (int k = 0; k < KH; ++k) qrow[k] = rc.rowstat[row * KH + k];
#pragma unroll
            for (int k = 0; k < KH; ++k) sv[k] = rc.s2[(int64_t)c * rc.lds2 + k];
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const float raw = qrow[k].x + sv[k];
                float r = raw;
                if (rc.mode == 0) r = r > 0.f ? r : r * rc.slope;
                w[k] = (expf(r - qrow[k].y) - qrow[k].w) * qrow[k].z;
                ga[k] = rc.mode == 0 ? w[k] * (raw > 0.f ? 1.f : rc.slope) : w[k] + qrow[k].w * qrow[k].z;
            }
        } else if constexpr (RECOMPUTE) {
            const float4 *st = rc.rowstat + (int64_t)c * KH;
            float4 q[KH];
#pragma unroll
            for (int k = 0; k < KH; ++k) q[k] = st[k];
            if constexpr (GRAD) {
#pragma unroll
                for (int k = 0; k < KH; ++k) gt[k] = eg.t[(int64_t)c * KH + k];
            }
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const float raw = q[k].x + a2[k];
                float r = raw;
                if (rc.mode == 0) r = r > 0.f ? r : r * rc.slope;
                w[k] = (expf(r - q[k].y) - q[k].w) * q[k].z;
                if constexpr (GRAD) {
                    float g = w[k] + (rc.mode == 1 ? q[k].w * q[k].z : 0.f);
                    if (rc.mode == 0) g *= raw > 0.f ? 1.f : rc.slope;
                    ga[k] = g;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KH; ++k) w[k] = __builtin_nontemporal_load(alpha + (int64_t)k * plane + idx);
        }
    };
    // prefetch of the first 64 (col, alpha) tuples: unconditional, index clamped into the task
    int32_t nc = 0;
    float na[KH], nga[KH], ngt[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) na[k] = nga[k] = ngt[k] = 0.f;
    if (len > 0) {
        const int e = lane < last ? lane : last;
        nc = __builtin_nontemporal_load(col + kbeg + e);
        weights(kbeg + e, nc, na, nga, ngt);
        if constexpr (FWD) {
#pragma unroll
            for (int k = 0; k < KH; ++k) csum[k] += lane < len ? nga[k] : 0.f;
        }
    }
    for (int base = 0; base < len; base += 64) {   // wave-uniform: one task per wave
        __builtin_amdgcn_wave_barrier();
        mine[lane * PS] = __int_as_float(nc);
#pragma unroll
        for (int k = 0; k < KH; ++k) mine[lane * PS + 1 + k] = na[k];
        if constexpr (GRAD) {
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                mine[lane * PS + 1 + KH + k] = nga[k];
                mine[lane * PS + 1 + 2 * KH + k] = ngt[k];
            }
        }
        if constexpr (FWD) {
#pragma unroll
            for (int k = 0; k < KH; ++k) mine[lane * PS + 1 + KH + k] = nga[k];
        }
        __builtin_amdgcn_wave_barrier();
        if (base + 64 < len) {
            int e = base + 64 + lane;
            e = e < last ? e : last;
            nc = __builtin_nontemporal_load(col + kbeg + e);
            weights(kbeg + e, nc, na, nga, ngt);
            if constexpr (FWD) {
#pragma unroll
                for (int k = 0; k < KH; ++k) csum[k] += base + 64 + lane < len ? nga[k] : 0.f;
            }
        }
        const int cnt = min(64, len - base);
        // GRAD: the dot products of a batch are FINISHED (16-lane butterfly, de, row sum) one batch later, after the
        // next batch's gathers have gone out: the reduction's chain of cross-lane round trips runs under their latency
        float dq[kBatch];
        int pe0 = -1;                          // first entry of the batch whose dot products are pending
        auto finish = [&](int e0p) {
            float dm;
            if (hl == 16) {
                // 8 entries x 16 lanes: a transposing reduction on the DPP network (no LDS round trips) -- each step
                // halves the entries a lane carries while it doubles the lanes summed: 26 VALU instructions instead
                // of 32 ds_bpermute + 32 adds.  Lane l of the head ends with entry l >> 1.
                const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
                float r1[4], r2[2];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    r1[u] = (b3 ? dq[u + 4] : dq[u]) + dpp_f<0x128>(b3 ? dq[u] : dq[u + 4]);          // row_ror:8 = lane ^ 8
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    r2[u] = (b2 ? r1[u + 2] : r1[u]) + dpp_xor4(b2 ? r1[u] : r1[u + 2]);
                const float r3 = (b1 ? r2[1] : r2[0]) + dpp_f<0x4E>(b1 ? r2[0] : r2[1]);               // quad_perm [2,3,0,1]
                dm = r3 + dpp_f<0xB1>(r3);                                                            // quad_perm [1,0,3,2]
            } else {
                for (int o = hl >> 1; o > 0; o >>= 1) {
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) dq[u] += __shfl_xor(dq[u], o, 64);
                }
                dm = dq[0];
#pragma unroll
                for (int u = 1; u < kBatch; ++u) dm = u_mine == u ? dq[u] : dm;
            }
            const int e = e0p + u_mine;        // the entry of the batch this lane finishes
            if (fact && writer && e < cnt) {
                const float g = mine[e * PS + 1 + KH + head] * (dm - mine[e * PS + 1 + 2 * KH + head]);
                if (eg.de) eg.de[(kbeg + base + e) * KH + head] = g;
                acc2 += g;
            }
        };
        for (int e0 = 0; e0 < cnt; e0 += kBatch) {
            int32_t c[kBatch];
            float w[kBatch], w2[kBatch];
            float x[kBatch][4];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int e = e0 + u < cnt ? e0 + u : cnt - 1;       // ragged tail: the task's last referenced row
                c[u] = __float_as_int(mine[e * PS]);
                w[u] = mine[e * PS + 1 + head];
                w2[u] = FWD ? mine[e * PS + 1 + KH + head] : 0.f;
            }
            // (lanes beyond the F features gather column 0 and are never stored: no branch around the gathers, so the
            //  rows stay in flight across the pending reduction below instead of being copied out of a conditional)
#pragma unroll
            for (int u = 0; u < kBatch; ++u) vload<4>(x[u], B + (int64_t)c[u] * ldb + fsafe);
            if constexpr (GRAD) {
                __builtin_amdgcn_sched_barrier(0);     // the gathers are out; nothing below may be moved above them,
                if (pe0 >= 0) finish(pe0);             // and no use of a gathered row above the pending reduction
#pragma unroll
                for (int u = 0; u < kBatch; ++u) dq[u] = 0.f;
                pe0 = e0;
                __builtin_amdgcn_sched_barrier(0);
            }
            // The kernel issues ~4 VALU cycles per instruction and wave: at 1 KB per entry it is as close to the issue
            // limit as to the gather limit.  Full batches (all but a task's last) take packed FMAs and no selects.
            if (e0 + kBatch <= cnt) {
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const f32x2 ww = {w[u], w[u]};
                    const f32x2 xl = {x[u][0], x[u][1]}, xh = {x[u][2], x[u][3]};
                    accl = __builtin_elementwise_fma(ww, xl, accl);
                    acch = __builtin_elementwise_fma(ww, xh, acch);
                    if constexpr (FWD) {
                        const f32x2 w22 = {w2[u], w2[u]};
                        vl = __builtin_elementwise_fma(w22, xl, vl);
                        vh = __builtin_elementwise_fma(w22, xh, vh);
                    }
                    if constexpr (GRAD) {
                        const f32x2 pr = __builtin_elementwise_fma(xh, zh, xl * zl);
                        dq[u] = pr.x + pr.y;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const bool keep = e0 + u < cnt;
                    const float wk = keep ? w[u] : 0.f;
                    const f32x2 ww = {wk, wk};
                    const f32x2 xl = {keep ? x[u][0] : 0.f, keep ? x[u][1] : 0.f}, xh = {keep ? x[u][2] : 0.f, keep ? x[u][3] : 0.f};
                    accl = __builtin_elementwise_fma(ww, xl, accl);
                    acch = __builtin_elementwise_fma(ww, xh, acch);
                    if constexpr (FWD) {
                        const float w2k = keep ? w2[u] : 0.f;
                        const f32x2 w22 = {w2k, w2k};
                        vl = __builtin_elementwise_fma(w22, xl, vl);
                        vh = __builtin_elementwise_fma(w22, xh, vh);
                    }
                    if constexpr (GRAD) {
                        const f32x2 pr = __builtin_elementwise_fma(xh, zh, xl * zl);
                        dq[u] = pr.x + pr.y;
                    }
                }
            }
        }
        if constexpr (GRAD) {
            if (pe0 >= 0) finish(pe0);         // (before the next 64 entries overwrite the parked words)
        }
    }
    const int64_t pw = GRAD ? (int64_t)eg.pw : (int64_t)F;     // width of a partial row
    if constexpr (GRAD) {
        // ds2 of the row = sum of this task's de per head: lane L < pw - F writes column F + L (pad columns: zero)
        for (int o = hl >> 1; o > 0; o >>= 1) acc2 += __shfl_xor(acc2, o, 64);
        const int src = (lane < KH ? lane : KH - 1) * hl;
        const float v = __shfl(acc2, src, 64);
        if (tact && lane < eg.pw - F) {
            float out = lane < KH ? v : 0.f;
            if (dst >= 0) {
                partial[(int64_t)dst * pw + F + lane] = out;
            } else {
                float *cp = C + (int64_t)(~dst) * ldc + F + lane;
                if (flags & PGCN_SPMM_ACCUMULATE) out += *cp;
                *cp = out;
            }
        }
    }
    if constexpr (FWD) {
        // V_i beside the output row; C_i = the wave's sum of c_ij per head: lane L < pw2 - F writes column F + L
        float mysum = 0.f;
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            float v = csum[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            mysum = lane == k ? v : mysum;
        }
        if (tact) {
            float *o2 = dst >= 0 ? f2.partial2 + (int64_t)dst * f2.pw2 : f2.C2 + (int64_t)(~dst) * f2.ldc2;
            const bool add = dst < 0 && (flags & PGCN_SPMM_ACCUMULATE);
            if (fact) {
                float v4[4] = {vl.x, vl.y, vh.x, vh.y};
                if (add) {
                    float old[4];
                    vload<4>(old, o2 + fcol);
#pragma unroll
                    for (int v = 0; v < 4; ++v) v4[v] += old[v];
                }
                vstore<4>(o2 + fcol, v4);
            }
            if (lane < f2.pw2 - F) {
                float out = lane < KH ? mysum : 0.f;
                if (add) out += o2[F + lane];
                o2[F + lane] = out;
            }
        }
    }
    float acc[4] = {accl.x, accl.y, acch.x, acch.y};
    if (tact && fact) {
        if (dst >= 0) {
            vstore<4>(partial + (int64_t)dst * pw + fcol, acc);
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
                 const Recompute *rc, const EdgeGrad *eg, const Forward2 *f2, int32_t heads, int32_t d, int64_t nrows, const int32_t *tasks, int64_t ntasks,
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
    int64_t pw = F;
    if (eg) {
        const int hl = d / 4;
        if (!rc || !eg->Z || !eg->t || eg->ldz < F || eg->ldz % 4 || (uintptr_t)eg->Z % 16)     // (de may be NULL: not kept)
            return pgcn_set_error2(PGCN_EINVAL, who, "bad Z / t");
        if (hl < kBatch || (hl & (hl - 1)))
            return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "the fused edge gradient needs d = 32, 64, 128 or 256");
        pw = eg->pw;
        if (pw != F + (heads + 3) / 4 * 4 || ldc < pw)
            return pgcn_set_error2(PGCN_EINVAL, who, "C must hold heads * d + heads (rounded up to 4) columns");
    }
    int64_t pw_all = pw;                 // floats of work-space per slot
    Forward2 h0{};
    if (f2) {
        const int hl = d / 4;
        if (!rc || eg || !f2->C2 || f2->pw2 != F + (heads + 3) / 4 * 4 || f2->ldc2 < f2->pw2 || f2->ldc2 % 4 || (uintptr_t)f2->C2 % 16)
            return pgcn_set_error2(PGCN_EINVAL, who, "bad second output (heads * d + heads columns, rounded up to 4)");
        if (hl < kBatch || (hl & (hl - 1)))
            return pgcn_set_error2(PGCN_EUNSUPPORTED, who, "the two-accumulator product needs d = 32, 64, 128 or 256");
        pw_all = F + f2->pw2;
        h0 = *f2;
        h0.partial2 = partial_ws ? partial_ws + nslots * F : nullptr;
    }
    if (nslots < 0 || partial_ws_elems < nslots * pw_all)
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
    EdgeGrad g0{};
    if (eg) g0 = *eg;
#define PGCN_HEADS_LAUNCH(KH, RC, GR, FW)                                                                            \
    hipLaunchKernelGGL((spmm_heads_kernel<KH, RC, GR, FW>), dim3((unsigned)grid), dim3(kThreads), 0, s, rowptr, col,    \
                       alpha, plane_stride, t4, nt, B, ldb, C, ldc, (int32_t)F, d, partial_ws, flags, nslices, sg, r0,  \
                       g0, h0)
#define PGCN_HEADS(KH)                                                                                              \
    case KH:                                                                                                        \
        if (f2) PGCN_HEADS_LAUNCH(KH, true, false, true);                                                            \
        else if (eg) PGCN_HEADS_LAUNCH(KH, true, true, false);                                                       \
        else if (rc) PGCN_HEADS_LAUNCH(KH, true, false, false);                                                      \
        else PGCN_HEADS_LAUNCH(KH, false, false, false);                                                             \
        break;
    switch (heads) {
        PGCN_HEADS(1) PGCN_HEADS(2) PGCN_HEADS(3) PGCN_HEADS(4) PGCN_HEADS(5) PGCN_HEADS(6) PGCN_HEADS(7) PGCN_HEADS(8)
    }
#undef PGCN_HEADS
#undef PGCN_HEADS_LAUNCH
    PGCN_HIP_CHECK(hipGetLastError());
    if (nfix > 0) {    // (GRAD: the ds2 columns of a split row are combined with its features, one list, one launch)
        const int rcf = pgcn_spmm_fixup_f32(fix, nfix, nullptr, nullptr, partial_ws, C, ldc, (int32_t)pw,
                                            flags & PGCN_SPMM_ACCUMULATE, stream);
        if (rcf != PGCN_OK || !f2) return rcf;
        return pgcn_spmm_fixup_f32(fix, nfix, nullptr, nullptr, h0.partial2, h0.C2, h0.ldc2, h0.pw2,
                                   flags & PGCN_SPMM_ACCUMULATE, stream);
    }
    return PGCN_OK;
}
}  // namespace

extern "C" int pgcn_spmm_heads_f32(const int64_t *rowptr, const int32_t *col, const float *alpha, int64_t plane_stride,
                                   int32_t heads, int32_t d, int64_t nrows, const int32_t *tasks, int64_t ntasks,
                                   const int64_t *seg, int32_t nslices, const int32_t *fix, int64_t nfix,
                                   const float *B, int64_t ldb, float *C, int64_t ldc, float *partial_ws,
                                   int64_t partial_ws_elems, int64_t nslots, uint32_t flags, pgcn_stream_t stream) {
    return launch_heads("pgcn_spmm_heads_f32", rowptr, col, alpha, plane_stride, nullptr, nullptr, nullptr, heads, d, nrows, tasks, ntasks, seg,
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
    return launch_heads("pgcn_spmm_heads_recompute_f32", rowptr, col, nullptr, 0, &rc, nullptr, nullptr, heads, d, nrows, tasks, ntasks,
                        seg, nslices, fix, nfix, B, ldb, C, ldc, partial_ws, partial_ws_elems, nslots, flags, stream);
}

extern "C" int pgcn_spmm_heads_grad_f32(const int64_t *rowptr, const int32_t *col, const float *rowstat, const float *s2,
                                        int64_t lds2, float slope, int32_t mode, int32_t heads, int32_t d, int64_t nrows,
                                        const int32_t *tasks, int64_t ntasks, const int64_t *seg, int32_t nslices,
                                        const int32_t *fix, int64_t nfix, const float *B, int64_t ldb, const float *Z,
                                        int64_t ldz, const float *t, float *C, int64_t ldc, float *de, float *partial_ws,
                                        int64_t partial_ws_elems, int64_t nslots, uint32_t flags, pgcn_stream_t stream) {
    const Recompute rc{reinterpret_cast<const float4 *>(rowstat), s2, lds2, nrows, slope, mode};
    const EdgeGrad eg{Z, ldz, t, de, heads * d + (heads + 3) / 4 * 4};
    return launch_heads("pgcn_spmm_heads_grad_f32", rowptr, col, nullptr, 0, &rc, &eg, nullptr, heads, d, nrows, tasks, ntasks,
                        seg, nslices, fix, nfix, B, ldb, C, ldc, partial_ws, partial_ws_elems, nslots, flags, stream);
}

extern "C" int pgcn_spmm_heads_forward2_f32(const int64_t *rowptr, const int32_t *col, const float *rowstat, const float *s2,
                                            int64_t lds2, float slope, int32_t mode, int32_t heads, int32_t d, int64_t nrows,
                                            const int32_t *tasks, int64_t ntasks, const int64_t *seg, int32_t nslices,
                                            const int32_t *fix, int64_t nfix, const float *B, int64_t ldb, float *C,
                                            int64_t ldc, float *C2, int64_t ldc2, float *partial_ws,
                                            int64_t partial_ws_elems, int64_t nslots, uint32_t flags, pgcn_stream_t stream) {
    const Recompute rc{reinterpret_cast<const float4 *>(rowstat), s2, lds2, nrows, slope, mode};
    const Forward2 f2{C2, ldc2, nullptr, heads * d + (heads + 3) / 4 * 4};
    return launch_heads("pgcn_spmm_heads_forward2_f32", rowptr, col, nullptr, 0, &rc, nullptr, &f2, heads, d, nrows, tasks,
                        ntasks, seg, nslices, fix, nfix, B, ldb, C, ldc, partial_ws, partial_ws_elems, nslots, flags, stream);
}
