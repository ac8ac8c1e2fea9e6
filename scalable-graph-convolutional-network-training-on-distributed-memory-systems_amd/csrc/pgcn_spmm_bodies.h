// pgcn_spmm_bodies.h -- the two SpMM work-item bodies as device functions: pgcn_spmm.hip runs the gather tasks,
// pgcn_spmm_core.hip the LDS-tiled core pieces (a third kernel that co-scheduled both kinds on one CU was measured
// in r01 -- no gain -- and lives in tools/experiments/pgcn_spmm_fused.hip, outside the product build).
#ifndef PGCN_SPMM_BODIES_H
#define PGCN_SPMM_BODIES_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgcn_device.h"
#include "pgcn_internal.h"

namespace pgcn_bodies {

constexpr int kUnroll = 8;              // row gathers in flight per batch (gather body)
constexpr int TR = PGCN_CORE_TR;        // rows per tile (core body)
constexpr int TC = PGCN_CORE_TC;        // columns per panel
constexpr int kCoreThreads = 512;
constexpr int NG = kCoreThreads / 32;   // groups per workgroup
constexpr int RW = TR / NG;             // rows per group
static_assert(TR % NG == 0, "tile rows must divide over the groups");
#ifndef PGCN_CORE_BATCH
#define PGCN_CORE_BATCH 4
#endif
constexpr int CB = PGCN_CORE_BATCH;     // LDS row reads in flight per batch
#ifndef PGCN_CORE_STAGE_BATCH
#define PGCN_CORE_STAGE_BATCH 4
#endif
constexpr size_t core_smem_bytes(int vec) { return (size_t)(TC + 1) * 32 * vec * 4 + (kCoreThreads / 64) * 512; }

template <int VEC>
__device__ __forceinline__ void vfma(float (&acc)[VEC], float w, const float (&x)[VEC]) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, x[v], acc[v]);
}

// Row load of B.  OFF32: the caller guarantees that every byte offset into B fits 32 bits,
// so the address is (uniform 64-bit base) + (32-bit lane offset): one v_mul_lo + v_add per
// load and an SGPR base instead of a 64-bit multiply-add and two address VGPRs per load.
template <int VEC, bool OFF32>
__device__ __forceinline__ void load_row(float (&x)[VEC], const float *B, uint32_t lane_byte_off,
                                         int32_t c, int64_t ldb) {
    const float *p;
    if constexpr (OFF32) {
        const uint32_t off = (uint32_t)c * (uint32_t)(ldb * 4) + lane_byte_off;
        p = reinterpret_cast<const float *>(reinterpret_cast<const char *>(B) + off);
    } else {
        p = B + (int64_t)c * ldb + (lane_byte_off >> 2);
    }
    vload<VEC>(x, p);   // default cache policy: non-temporal gathers were measured 1.7x slower (r01)
}


// ---------------------------------------------------------------------------------------------
// Gather task: one group of LPR lanes accumulates `len` stored entries starting at kbeg into its
// VEC features and writes a partial-sum slot (dst >= 0) or row ~dst of C.  `mrow` = 64 float2 of
// LDS owned by the calling wave.  See pgcn_spmm.hip for the design notes.
template <int LPR, int VEC, bool HAS_VAL, bool OFF32>
__device__ __forceinline__ void gather_task_body(
    const bool tact, const int64_t kbeg, const int32_t len, const int32_t dst,
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const float *__restrict__ val,
    const int32_t *__restrict__ row_map, const float *__restrict__ B, const int64_t ldb,
    float *__restrict__ C, const int64_t ldc, const int32_t f, float *__restrict__ partial,
    const uint32_t flags, const int fcol, float2 *mrow) {
    constexpr int U = (LPR < kUnroll) ? LPR : kUnroll;   // gathers in flight per batch
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int gbase = (lane / LPR) * LPR;
    const bool fact = fcol < f;
    const uint32_t lane_off = (uint32_t)fcol * 4u;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

    // (col,val) pairs: one per lane, streamed once -> non-temporal.  The loads are
    // UNCONDITIONAL (index clamped into the task; an empty / inactive group reads a
    // harmless valid word of rowptr) so that the prefetch of the next LPR pairs stays in
    // flight behind the gathers instead of being fenced at a branch join.
    const int32_t *cp = (len > 0) ? col + kbeg : reinterpret_cast<const int32_t *>(rowptr);
    const float *vp = (len > 0) ? val + kbeg : reinterpret_cast<const float *>(rowptr);
    const int last = (len > 0) ? len - 1 : 0;
    int32_t nc;
    float nv = 1.f;
    {
        const int e = (sub < last) ? sub : last;
        nc = __builtin_nontemporal_load(cp + e);
        if constexpr (HAS_VAL) nv = __builtin_nontemporal_load(vp + e);
    }
    // The wave's current LPR-batch of pairs lives in LDS (512 B per wave): a group
    // broadcasts entry k to its lanes with ONE ds_read_b128 per two entries (all lanes of
    // a group read the same address = conflict-free broadcast), instead of two
    // ds_bpermute per entry -- the LDS pipe is shared by the whole CU and was the
    // co-bottleneck of the gather loop.  Written and read by the same wave only, LDS
    // operations of a wave execute in order => no workgroup barrier.
    for (int base = 0; __any(base < len); base += LPR) {
        __builtin_amdgcn_wave_barrier();
        mrow[lane] = make_float2(__int_as_float(len > 0 ? nc : 0), nv);
        __builtin_amdgcn_wave_barrier();
        {
            int e = base + LPR + sub;
            e = (e < last) ? e : last;
            nc = __builtin_nontemporal_load(cp + e);
            if constexpr (HAS_VAL) nv = __builtin_nontemporal_load(vp + e);
        }
        const int cnt = len - base;  // entries left for this group (may be <= 0)
        const float2 *mg = mrow + gbase;
#pragma unroll
        for (int k = 0; k < LPR; k += U) {
            if (!__any(k < cnt)) break;
            // U independent row loads, all unpredicated and issued back to back (a branch per
            // load serialises them behind the broadcasts).  In a ragged batch the surplus
            // slots re-read the task's LAST referenced row (the clamped pair parked above; row 0
            // for an empty group) and are zeroed by a select before the FMA, so no row the task
            // does not reference is ever combined into the result (no 0 * Inf).
            const bool full = __all(k + U <= cnt);
            int32_t c[U];
            float w[U];
            float x[U][VEC];
            if constexpr (U >= 2) {
#pragma unroll
                for (int u = 0; u < U; u += 2) {
                    const float4 m = *reinterpret_cast<const float4 *>(mg + k + u);
                    c[u] = __float_as_int(m.x); w[u] = m.y;
                    c[u + 1] = __float_as_int(m.z); w[u + 1] = m.w;
                }
            } else {
                const float2 m = mg[k];
                c[0] = __float_as_int(m.x); w[0] = m.y;
            }
            if (fact) {
#pragma unroll
                for (int u = 0; u < U; ++u) load_row<VEC, OFF32>(x[u], B, lane_off, c[u], ldb);
                if (!full) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool keep = k + u < cnt;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) x[u][v] = keep ? x[u][v] : 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) vfma<VEC>(acc, w[u], x[u]);
            }
        }
    }

    if (tact && fact) {
        if (dst >= 0) {
            vstore<VEC>(partial + (int64_t)dst * f + fcol, acc);
        } else {
            const int32_t row = ~dst;
            const int64_t orow = row_map ? row_map[row] : row;
            float *c = C + orow * ldc + fcol;
            if (flags & PGCN_SPMM_ACCUMULATE) {
                float old[VEC];
                vload<VEC>(old, c);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += old[v];
            }
            vstore<VEC>(c, acc);
        }
    }
}

template <int VEC>
__device__ __forceinline__ void lds_row(float (&x)[VEC], const float *panel, int c, int sub) {
    vload<VEC>(x, panel + (c * 32 + sub) * VEC);
}

// ---------------------------------------------------------------------------------------------
// Core piece: one 512-thread workgroup, one row tile x a run of dense column panels staged in LDS.
// wk = {tile row index, first dense tile, one-past-last dense tile, first slot}; smem = dynamic LDS of
// core_smem_bytes(VEC).  See pgcn_spmm_core.hip for the design notes.
template <int VEC>
__device__ __forceinline__ void core_piece_body(
    const int4 wk, const int32_t *__restrict__ tile_panel, const int64_t *__restrict__ tile_base,
    const int32_t *__restrict__ seg_off, const int32_t *__restrict__ ccol, const float *__restrict__ cval,
    const float *__restrict__ B, const int64_t ldb, const int64_t ncols, const int32_t f,
    float *__restrict__ partial, char *smem, const int fcol0) {
    float *panel = reinterpret_cast<float *>(smem);                                        // (TC+1) x 32 x VEC
    float2 *mpark = reinterpret_cast<float2 *>(smem + (size_t)(TC + 1) * 32 * VEC * 4);     // per wave 64 pairs

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & 31;
    const int gbase = lane & 32;
    const int group = threadIdx.x >> 5;
    const int fcol = fcol0 + sub * VEC;
    const bool fact = fcol < f;
    float2 *mrow = mpark + wave * 64;
    const float2 *mg = mrow + gbase;

    float acc[RW][VEC];
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] = 0.f;

    if (threadIdx.x < 32) {   // the all-zero padding row
        float z[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] = 0.f;
        vstore<VEC>(panel + ((size_t)TC * 32 + threadIdx.x) * VEC, z);
    }

    for (int k = wk.y; k < wk.z; ++k) {
        const int64_t col0 = (int64_t)tile_panel[k] * TC;
        const int64_t base = tile_base[k];
        const int32_t *cb = ccol + base;   // uniform per workgroup
        const float *vb = cval + base;
        int32_t bound[RW + 1];
        {
            const int32_t *so = seg_off + (int64_t)k * (TR + 1) + group * RW;
#pragma unroll
            for (int j = 0; j <= RW; ++j) bound[j] = so[j];
        }
        // (1) first batch of every segment of this group
        int32_t pc[RW];
        float pv[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int len = bound[j + 1] - bound[j];
            const int e = (sub < len - 1) ? sub : len - 1;
            const uint32_t idx = (len > 0) ? (uint32_t)(bound[j] + e) : 0u;   // 32-bit lane offset, uniform base
            pc[j] = __builtin_nontemporal_load(cb + idx);
            pv[j] = __builtin_nontemporal_load(vb + idx);
        }
        __syncthreads();   // everyone is done reading the previous panel
        // stage B[col0 .. col0+TC) x [fcol0 .. fcol0+32*VEC) : one contiguous-row copy.  All
        // loads are issued before the first LDS write (clamped addresses, no branches: a
        // per-iteration bounds branch serialises the eight round trips); out-of-range
        // rows / features are zeroed by a select.
        {
            constexpr int NIT = TC * 32 / kCoreThreads;   // 8 row-vectors per thread
            constexpr int HB = PGCN_CORE_STAGE_BATCH;     // loads in flight per thread
            const int64_t lastrow = ncols - 1;
            const int lastf = f - VEC;
#pragma unroll
            for (int h = 0; h < NIT; h += HB) {
                float xs[HB][VEC];
#pragma unroll
                for (int it = 0; it < HB; ++it) {
                    const int idx = (h + it) * kCoreThreads + threadIdx.x;
                    const int r = idx >> 5, s = idx & 31;
                    const int64_t rr = (col0 + r < ncols) ? col0 + r : lastrow;
                    const int cc = (fcol0 + s * VEC < f) ? fcol0 + s * VEC : lastf;
                    vload<VEC>(xs[it], B + rr * ldb + cc);
                }
#pragma unroll
                for (int it = 0; it < HB; ++it) {
                    const int idx = (h + it) * kCoreThreads + threadIdx.x;
                    const int r = idx >> 5, s = idx & 31;
                    const bool ok = (col0 + r < ncols) && (fcol0 + s * VEC < f);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) xs[it][v] = ok ? xs[it][v] : 0.f;
                    vstore<VEC>(panel + (size_t)idx * VEC, xs[it]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int len = bound[j + 1] - bound[j];
            int32_t cur_c = pc[j];
            float cur_v = pv[j];
            for (int b = 0; __any(b < len); b += 32) {
                const int cnt = len - b;   // entries left in this group's segment (may be <= 0)
                int32_t nx_c = 0;
                float nx_v = 0.f;
                if (__any(b + 32 < len)) {   // (2) one batch ahead
                    int e = b + 32 + sub;
                    e = (e < len - 1) ? e : len - 1;
                    const uint32_t idx = (len > 0) ? (uint32_t)(bound[j] + e) : 0u;
                    nx_c = __builtin_nontemporal_load(cb + idx);
                    nx_v = __builtin_nontemporal_load(vb + idx);
                }
                const bool valid = sub < cnt;
                __builtin_amdgcn_wave_barrier();
                mrow[lane] = make_float2(__int_as_float(valid ? cur_c : TC), valid ? cur_v : 0.f);
                __builtin_amdgcn_wave_barrier();
                // software-pipelined: the pairs of batch k4+CB are read while batch k4 computes, so a
                // batch costs one LDS round trip (rows) instead of two (pairs, then rows).
                float4 m[CB / 2];
#pragma unroll
                for (int u = 0; u < CB / 2; ++u) m[u] = *reinterpret_cast<const float4 *>(mg + 2 * u);
#pragma unroll
                for (int k4 = 0; k4 < 32; k4 += CB) {
                    if (!__any(k4 < cnt)) break;
                    float4 mn[CB / 2];
                    if (k4 + CB < 32) {
#pragma unroll
                        for (int u = 0; u < CB / 2; ++u) mn[u] = *reinterpret_cast<const float4 *>(mg + k4 + CB + 2 * u);
                    }
                    float x[CB][VEC];
#pragma unroll
                    for (int u = 0; u < CB / 2; ++u) {
                        lds_row<VEC>(x[2 * u], panel, __float_as_int(m[u].x), sub);
                        lds_row<VEC>(x[2 * u + 1], panel, __float_as_int(m[u].z), sub);
                    }
#pragma unroll
                    for (int u = 0; u < CB / 2; ++u) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[j][v] = fmaf(m[u].y, x[2 * u][v], acc[j][v]);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[j][v] = fmaf(m[u].w, x[2 * u + 1][v], acc[j][v]);
                    }
                    if (k4 + CB < 32) {
#pragma unroll
                        for (int u = 0; u < CB / 2; ++u) m[u] = mn[u];
                    }
                }
                cur_c = nx_c;
                cur_v = nx_v;
            }
        }
    }
    if (fact) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int rit = j * NG + group;   // row inside the tile
            vstore<VEC>(partial + ((int64_t)wk.w + rit) * f + fcol, acc[j]);
        }
    }
}

}  // namespace pgcn_bodies
#endif
