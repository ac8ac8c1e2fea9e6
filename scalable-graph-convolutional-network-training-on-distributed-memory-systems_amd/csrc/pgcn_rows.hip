// pgcn_rows.hip -- boundary-row pack / unpack for gfx950.
//
//   gather : out[r,:]      = H[idx[r],:]     replaces H[indices]        GPU/PGCN.py:104
//   scatter: H[idx[r],:] (+)= in[r,:]        replaces X[indices] = buf  GPU/PGCN.py:115
//                                            (+= : accumulate-on-receive, main.c:295,400)
//
// Pure HBM-bound row copies: a group of LPR lanes moves one row as 16-byte
// vectors (fully coalesced on the contiguous side, row-granular on the indexed
// side); four independent rows per group iteration keep loads in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgcn_internal.h"

namespace {

constexpr int kThreads = 256;

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

template <int VEC, bool GATHER, bool ACC>
__global__ __launch_bounds__(kThreads) void rows_kernel(
    float *__restrict__ indexed, int64_t ld_indexed, const int32_t *__restrict__ idx,
    int64_t nidx, float *__restrict__ packed, int64_t ld_packed, int32_t nvec, int32_t lpr) {
    using V = typename VecT<VEC>::type;
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t grp = t / lpr;
    const int sub = (int)(t % lpr);
    const int64_t ngrp = ((int64_t)gridDim.x * kThreads) / lpr;
    for (int64_t r = grp; r < nidx; r += ngrp) {
        const int64_t row = idx[r];
        V *pi = reinterpret_cast<V *>(indexed + row * ld_indexed);
        V *pp = reinterpret_cast<V *>(packed + r * ld_packed);
        for (int v = sub; v < nvec; v += lpr) {
            if constexpr (GATHER) {
                pp[v] = pi[v];
            } else if constexpr (!ACC) {
                pi[v] = pp[v];
            } else {
                // accumulate: an index may occur several times (a boundary row that comes back from several
                // peers) -- atomic adds, so no partial sum is lost (the order of the adds is then not fixed; the
                // engine's own reverse exchange uses a deterministic pattern SpMM instead, partition.unpack)
                const V b = pp[v];
                float *pf = reinterpret_cast<float *>(pi + v);
                if constexpr (VEC == 1) { atomicAdd(pf, b); }
                else { atomicAdd(pf, b.x); atomicAdd(pf + 1, b.y); atomicAdd(pf + 2, b.z); atomicAdd(pf + 3, b.w); }
            }
        }
    }
}

template <bool GATHER, bool ACC>
int launch_rows(float *indexed, int64_t ld_indexed, const int32_t *idx, int64_t nidx, float *packed,
                int64_t ld_packed, int32_t f, hipStream_t s) {
    const bool al16 = ((uintptr_t)indexed % 16 == 0) && ((uintptr_t)packed % 16 == 0) &&
                      ld_indexed % 4 == 0 && ld_packed % 4 == 0 && f % 4 == 0;
    const int vec = al16 ? 4 : 1;
    const int nvec = f / vec;
    int lpr = 1;
    while (lpr < nvec && lpr < 64) lpr *= 2;
    const int64_t groups_per_block = kThreads / lpr;
    int64_t grid = (nidx + groups_per_block - 1) / groups_per_block;
    if (grid > 256 * 16) grid = 256 * 16;  // grid-stride beyond 16 blocks per CU
    if (grid < 1) grid = 1;
    if (vec == 4)
        hipLaunchKernelGGL((rows_kernel<4, GATHER, ACC>), dim3((unsigned)grid), dim3(kThreads), 0, s,
                           indexed, ld_indexed, idx, nidx, packed, ld_packed, nvec, lpr);
    else
        hipLaunchKernelGGL((rows_kernel<1, GATHER, ACC>), dim3((unsigned)grid), dim3(kThreads), 0, s,
                           indexed, ld_indexed, idx, nidx, packed, ld_packed, nvec, lpr);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

}  // namespace

extern "C" int pgcn_gather_rows_f32(const float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                                    float *out, int64_t ldo, int32_t f, pgcn_stream_t stream) {
    if (nidx < 0 || f <= 0 || ldh < f || ldo < f)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_gather_rows_f32: bad sizes");
    if (nidx == 0) return PGCN_OK;
    if (!H || !idx || !out) return pgcn_set_error(PGCN_EINVAL, "pgcn_gather_rows_f32: null pointer");
    return launch_rows<true, false>(const_cast<float *>(H), ldh, idx, nidx, out, ldo, f,
                                    (hipStream_t)stream);
}

extern "C" int pgcn_scatter_rows_f32(float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                                     const float *in, int64_t ldi, int32_t f, int32_t accumulate,
                                     pgcn_stream_t stream) {
    if (nidx < 0 || f <= 0 || ldh < f || ldi < f)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_scatter_rows_f32: bad sizes");
    if (nidx == 0) return PGCN_OK;
    if (!H || !idx || !in) return pgcn_set_error(PGCN_EINVAL, "pgcn_scatter_rows_f32: null pointer");
    if (accumulate)
        return launch_rows<false, true>(H, ldh, idx, nidx, const_cast<float *>(in), ldi, f,
                                        (hipStream_t)stream);
    return launch_rows<false, false>(H, ldh, idx, nidx, const_cast<float *>(in), ldi, f,
                                     (hipStream_t)stream);
}
