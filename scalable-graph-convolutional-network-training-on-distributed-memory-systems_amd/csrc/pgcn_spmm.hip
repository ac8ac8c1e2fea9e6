// pgcn_spmm.hip -- CSR SpMM for gfx950 (MI355X / CDNA4), fp32.
//
//   C[nrows x f] (+)= A[nrows x *] . B[* x f]
//
// Replaces torch.sparse.mm at /root/reference/GPU/PGCN.py:127,132 (== GrB_mxm on
// PLUS_TIMES_FP32, Parallel-GCN/main.c:271,295,376,400).
//
// Mapping to the machine (wave = 64 lanes, no tensor-core reshaping: this is
// HBM / cache-bandwidth bound gather work):
//   * a GROUP of LPR lanes owns one task (= one row, or one <=chunk segment of
//     a long row); each lane of the group owns VEC consecutive features, so a
//     group reads one dense row of B as one fully coalesced LPR*VEC*4-byte
//     segment (f=128: 32 lanes x 16 B = 512 B; two tasks per wave make the
//     1 KiB global_load_dwordx4 the L1 likes).  G = 64/LPR tasks per wave.
//   * (col,val) pairs of a task are fetched LPR at a time, one pair per lane
//     (coalesced, non-temporal: they are streamed exactly once and must not
//     evict feature rows from the per-XCD L2), parked in 512 B of LDS per wave
//     and broadcast inside the group with ds_read_b128 (2 pairs per read).
//   * the gather loop is unrolled x8 so each wave keeps eight independent
//     row-loads in flight; fp32 FMA accumulation in registers; no atomics.
//   * rows longer than the plan's chunk are split; segment partial sums go to
//     a caller-provided work-space and are combined in fixed order by
//     spmm_fixup_kernel (deterministic).
//   * XCD swizzle: block b runs on XCD b%8 (private 4 MiB L2); the optional
//     remap hands each XCD one contiguous range of rows so neighbouring rows
//     (which share columns in any locality-preserving ordering) share an L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pgcn_spmm_bodies.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * 64;

__device__ __forceinline__ int64_t swizzle_block(int64_t b, int64_t nb, bool on) {
    if (!on) return b;
    // block b lands on XCD b % 8; give XCD x the contiguous block range
    // [x*per, (x+1)*per).  Bijective for any nb via the ceil-sized ranges +
    // bounds check in the caller (ids >= nb simply have no task).
    const int64_t per = (nb + 7) / 8;
    return (b % 8) * per + b / 8;
}

struct SliceSeg { int64_t v[PGCN_MAX_SLICES + 1]; };

// tasks: int4 {kbeg low 32, kbeg high 32, length, dst}; kbeg = absolute offset of the task's
// first entry in col/val; dst >= 0: partial-sum slot, dst < 0: write row ~dst of C directly.
template <int LPR, int VEC, bool HAS_VAL, bool OFF32>
__global__ __launch_bounds__(kThreads, 6) void spmm_tasks_kernel(
    const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, const int4 *__restrict__ tasks, int64_t ntasks,
    const int32_t *__restrict__ row_map, const float *__restrict__ B, int64_t ldb,
    float *__restrict__ C, int64_t ldc, int32_t f, float *__restrict__ partial,
    int64_t nblocks, uint32_t flags, int32_t nslices, SliceSeg seg) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane / LPR;
    const int sub = lane % LPR;

    __shared__ float2 meta_lds[kWavesPerBlock][64];
    // grid y walks the features LPR * VEC at a time (one tile for whole rows; two 64-feature passes with
    // PGCN_SPMM_FPASS64: dispatched pass-major, so a pass gathers 256-byte rows from half the working set)
    const int fcol = (blockIdx.y * LPR + sub) * VEC;  // first feature owned by this lane
    int64_t tid, nt = ntasks;
    if (nslices > 1) {
        // workgroup b runs on XCD b % 8: it takes tasks of slice b % nslices only, so
        // this XCD's L2 sees just the rows of B with (col % nslices) == slice.
        const int slice = blockIdx.x % nslices;
        const int64_t sb = blockIdx.x / nslices;
        tid = seg.v[slice] + (sb * kWavesPerBlock + wave) * G + grp;
        nt = seg.v[slice + 1];
        if (seg.v[slice] + sb * kWavesPerBlock * G >= nt) return;
    } else {
        const int64_t bid = swizzle_block(blockIdx.x, nblocks, (flags & PGCN_SPMM_XCD_SWIZZLE) != 0);
        tid = (bid * kWavesPerBlock + wave) * G + grp;
        if (bid * kWavesPerBlock * G >= nt) return;
    }
    const bool tact = tid < nt;

    int32_t len = 0, dst = -1;
    int64_t kbeg = 0;
    if (tact) {
        if (tasks) {
            const int4 t = tasks[tid];
            kbeg = (int64_t)(((uint64_t)(uint32_t)t.y << 32) | (uint32_t)t.x);
            len = t.z;
            dst = t.w;
        } else {  // one task per row
            kbeg = rowptr[tid];
            len = (int32_t)(rowptr[tid + 1] - kbeg);
            dst = ~(int32_t)tid;
        }
    }
    pgcn_bodies::gather_task_body<LPR, VEC, HAS_VAL, OFF32>(tact, kbeg, len, dst, rowptr, col, val, row_map, B, ldb,
                                                          C, ldc, f, partial, flags, fcol, meta_lds[wave]);
}

// fix: int4 {row, first slot, #segments, unused}; sums the segments of a split
// row in slot order -> deterministic.
template <int LPR, int VEC>
__global__ __launch_bounds__(kThreads) void spmm_fixup_kernel(
    const int4 *__restrict__ fix, int64_t nfix, const int32_t *__restrict__ row_map,
    const float *__restrict__ partial, float *__restrict__ C, int64_t ldc, int32_t f,
    uint32_t flags) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane / LPR;
    const int sub = lane % LPR;
    const int64_t id = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * G + grp;
    const int fcol = (blockIdx.y * LPR + sub) * VEC;
    if (id >= nfix || fcol >= f) return;
    const int4 t = fix[id];
    const int64_t orow = row_map ? row_map[t.x] : t.x;
    float *c = C + orow * ldc + fcol;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (flags & PGCN_SPMM_ACCUMULATE) vload<VEC>(acc, c);
    const float *p = partial + (int64_t)t.y * f + fcol;
    int s = 0;
    for (; s + 4 <= t.z; s += 4) {   // four independent loads in flight, summed in slot order
        float x0[VEC], x1[VEC], x2[VEC], x3[VEC];
        vload<VEC>(x0, p + (int64_t)(s + 0) * f);
        vload<VEC>(x1, p + (int64_t)(s + 1) * f);
        vload<VEC>(x2, p + (int64_t)(s + 2) * f);
        vload<VEC>(x3, p + (int64_t)(s + 3) * f);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = (((acc[v] + x0[v]) + x1[v]) + x2[v]) + x3[v];
    }
    for (; s < t.z; ++s) {
        float x[VEC];
        vload<VEC>(x, p + (int64_t)s * f);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += x[v];
    }
    vstore<VEC>(c, acc);
}

struct Shape { int lpr, vec; };

Shape pick_shape(int32_t f, const void *B, int64_t ldb, const void *C, int64_t ldc,
                 const void *ws) {
    const bool al16 = ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) &&
                      ((uintptr_t)ws % 16 == 0) && ldb % 4 == 0 && ldc % 4 == 0 && f % 4 == 0;
    const int vec = al16 ? 4 : 1;
    const int nv = (f + vec - 1) / vec;
    int lpr = 1;
    while (lpr < nv && lpr < 64) lpr *= 2;
    return {lpr, vec};
}

template <int LPR, int VEC, bool HAS_VAL, bool OFF32>
int launch_tasks(const int64_t *rowptr, const int32_t *col, const float *val, const int4 *tasks,
                 int64_t ntasks, const int32_t *row_map, const float *B, int64_t ldb, float *C,
                 int64_t ldc, int32_t f, float *partial, int64_t grid, int ntiles, uint32_t flags,
                 int nslices, const SliceSeg &seg, hipStream_t s) {
    hipLaunchKernelGGL((spmm_tasks_kernel<LPR, VEC, HAS_VAL, OFF32>), dim3((unsigned)grid, ntiles),
                       dim3(kThreads), 0, s, rowptr, col, val, tasks, ntasks, row_map, B, ldb, C, ldc,
                       f, partial, grid, flags, nslices, seg);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}

template <int LPR, int VEC>
int launch(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *tasks,
           int64_t ntasks, const int64_t *seg_host, int nslices, const int32_t *fix, int64_t nfix,
           const int32_t *row_map,
           const float *B, int64_t ldb, float *C, int64_t ldc, int32_t f, float *partial,
           uint32_t flags, hipStream_t s) {
    constexpr int G = 64 / LPR;
    const int ntiles = (f + LPR * VEC - 1) / (LPR * VEC);
    if (ntasks > 0) {
        const int64_t per_block = (int64_t)kWavesPerBlock * G;
        SliceSeg seg{};
        int64_t grid;
        if (nslices > 1) {
            int64_t longest = 0;
            for (int i = 0; i <= nslices; ++i) seg.v[i] = seg_host[i];
            for (int i = 0; i < nslices; ++i)
                if (seg.v[i + 1] - seg.v[i] > longest) longest = seg.v[i + 1] - seg.v[i];
            grid = ((longest + per_block - 1) / per_block) * nslices;
        } else {
            const int64_t nblocks = (ntasks + per_block - 1) / per_block;
            grid = nblocks;
            if (flags & PGCN_SPMM_XCD_SWIZZLE) grid = ((nblocks + 7) / 8) * 8;
        }
        if (grid > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "spmm: too many tasks for one launch");
        const int4 *t4 = reinterpret_cast<const int4 *>(tasks);
        const bool off32 = (flags & PGCN_SPMM_OFFSETS32) != 0;
        int rc;
#define PGCN_LT(HV, O32)                                                                        \
    rc = launch_tasks<LPR, VEC, HV, O32>(rowptr, col, val, t4, ntasks, row_map, B, ldb, C, ldc, f, \
                                         partial, grid, ntiles, flags, nslices, seg, s)
        if (val) { if (off32) PGCN_LT(true, true); else PGCN_LT(true, false); }
        else     { if (off32) PGCN_LT(false, true); else PGCN_LT(false, false); }
#undef PGCN_LT
        if (rc != PGCN_OK) return rc;
    }
    if (nfix > 0 && !(flags & PGCN_SPMM_NO_FIXUP)) {
        const int64_t per_block = (int64_t)kWavesPerBlock * G;
        const int64_t grid = (nfix + per_block - 1) / per_block;
        hipLaunchKernelGGL((spmm_fixup_kernel<LPR, VEC>), dim3((unsigned)grid, ntiles), dim3(kThreads),
                           0, s, reinterpret_cast<const int4 *>(fix), nfix, row_map, partial, C, ldc,
                           f, flags);
        PGCN_HIP_CHECK(hipGetLastError());
    }
    return PGCN_OK;
}

int dispatch(const int64_t *rowptr, const int32_t *col, const float *val, const int32_t *tasks,
             int64_t ntasks, const int64_t *seg, int nslices, const int32_t *fix, int64_t nfix,
             const int32_t *row_map,
             const float *B, int64_t ldb, float *C, int64_t ldc, int32_t f, float *partial,
             uint32_t flags, hipStream_t s) {
    Shape sh = pick_shape(f, B, ldb, C, ldc, partial);
    // feature passes: fewer lanes per task -> the grid's y dimension walks the features 64 at a time (pass-major
    // dispatch order), so the rows of B one pass touches are 256 B each and a pass's working set is half the panel
    // (32-feature passes and a 1-D pass-major grid were measured in r02 / r03: slower / no difference)
    if (sh.vec == 4 && (flags & PGCN_SPMM_FPASS64) && sh.lpr > 16) sh.lpr = 16;
#define PGCN_CASE(L, V)                                                                       \
    if (sh.lpr == L && sh.vec == V)                                                           \
        return launch<L, V>(rowptr, col, val, tasks, ntasks, seg, nslices, fix, nfix, row_map, \
                            B, ldb, C, ldc, f, partial, flags, s);
    PGCN_CASE(1, 4) PGCN_CASE(2, 4) PGCN_CASE(4, 4) PGCN_CASE(8, 4) PGCN_CASE(16, 4)
    PGCN_CASE(32, 4) PGCN_CASE(64, 4)
    PGCN_CASE(1, 1) PGCN_CASE(2, 1) PGCN_CASE(4, 1) PGCN_CASE(8, 1) PGCN_CASE(16, 1)
    PGCN_CASE(32, 1) PGCN_CASE(64, 1)
#undef PGCN_CASE
    return pgcn_set_error(PGCN_EINVAL, "spmm: no kernel shape");
}

}  // namespace

extern "C" int pgcn_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                 int64_t nrows, const float *B, int64_t ldb, float *C,
                                 int64_t ldc, int32_t f, uint32_t flags, pgcn_stream_t stream) {
    if (nrows < 0 || f <= 0 || ldb < f || ldc < f)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_f32: bad sizes");
    if (nrows == 0) return PGCN_OK;
    if (!rowptr || !col || !B || !C)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_f32: null pointer");
    if (nrows > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_f32: nrows >= 2^31");
    return dispatch(rowptr, col, val, nullptr, nrows, nullptr, 1, nullptr, 0, nullptr, B, ldb, C,
                    ldc, f, nullptr, flags, (hipStream_t)stream);
}

extern "C" int pgcn_spmm_csr_plan_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                                      const int32_t *tasks, int64_t ntasks, const int64_t *seg,
                                      int32_t nslices, const int32_t *fix, int64_t nfix,
                                      const int32_t *row_map, const float *B, int64_t ldb,
                                      float *C, int64_t ldc, int32_t f, float *partial_ws,
                                      int64_t partial_ws_elems, int64_t nslots, uint32_t flags,
                                      pgcn_stream_t stream) {
    if (ntasks < 0 || nfix < 0 || f <= 0 || ldb < f || ldc < f || nslices < 1 ||
        nslices > PGCN_MAX_SLICES || (nslices > 1 && !seg))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_plan_f32: bad sizes");
    if (nslices > 1 && (seg[0] != 0 || seg[nslices] != ntasks))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_plan_f32: seg does not cover the task list");
    if (ntasks == 0) return PGCN_OK;
    const bool own_fixup = nfix > 0 && !(flags & PGCN_SPMM_NO_FIXUP);
    if (!rowptr || !col || !B || !C || !tasks || (own_fixup && !fix) || (nslots > 0 && !partial_ws))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_csr_plan_f32: null pointer");
    if (nslots < 0 || (nslots > 0 && partial_ws_elems < nslots * (int64_t)f))
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_csr_plan_f32: partial work-space too small");
    return dispatch(rowptr, col, val, tasks, ntasks, seg, nslices, fix, nfix, row_map, B, ldb, C,
                    ldc, f, partial_ws, flags, (hipStream_t)stream);
}
