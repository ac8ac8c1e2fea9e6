// pgcn_spmm_fused.hip -- gather tasks and LDS-tiled core pieces of ONE SpMM in ONE launch.
//
// The two stand-alone kernels stress different pipes of a CU: the gather part is bound by the
// L1/L2 path (row gathers), the core part by the LDS pipe (rows served from staged panels).  Run
// back to back -- or on two streams: the dispatcher does not start the second grid before the
// first drains -- their times add up.  Here every workgroup (512 threads, 68 KB LDS) takes one
// entry of a unified work list, either 16 gather tasks or one core piece, and the list
// interleaves the two kinds per XCD, so a CU typically hosts one workgroup of each kind and the
// L2-bound and the LDS-bound phases overlap in time.  Workgroup b runs on XCD b % 8: entry b of
// the list is a gather block of slice b % 8 (XCD-sliced plan) or any core piece.
// Both bodies are the ones of the stand-alone kernels (pgcn_spmm_bodies.h): results are
// bit-identical to the three-launch path, the fix-up kernel is unchanged.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgcn_spmm_bodies.h"

namespace {
using namespace pgcn_bodies;

// work: int4 {kind, a, b, 0}; kind 0: gather tasks [a, a+b), b <= 16; kind 1: core piece a; kind 2: nothing
template <bool OFF32>
__global__ __launch_bounds__(kCoreThreads, 4) void spmm_fused_kernel(
    const int4 *__restrict__ work, const int32_t *__restrict__ col, const float *__restrict__ val,
    const int4 *__restrict__ tasks, const int32_t *__restrict__ row_map,
    const int4 *__restrict__ core_work, const int32_t *__restrict__ tile_panel,
    const int64_t *__restrict__ tile_base, const int32_t *__restrict__ seg_off,
    const int32_t *__restrict__ ccol, const float *__restrict__ cval, const float *__restrict__ B,
    int64_t ldb, int64_t ncols, float *__restrict__ C, int64_t ldc, int32_t f,
    float *__restrict__ partial, uint32_t flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int4 w = work[blockIdx.x];
    if (w.x == 1) {
        core_piece_body<4>(core_work[w.y], tile_panel, tile_base, seg_off, ccol, cval, B, ldb, ncols, f, partial,
                           smem, 0);
        return;
    }
    if (w.x != 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot = wave * 2 + (lane >> 5);          // 16 groups of 32 lanes
    const bool tact = slot < w.z;
    int32_t len = 0, dst = -1;
    int64_t kbeg = 0;
    if (tact) {
        const int4 t = tasks[w.y + slot];
        kbeg = (int64_t)(((uint64_t)(uint32_t)t.y << 32) | (uint32_t)t.x);
        len = t.z;
        dst = t.w;
    }
    float2 *mrow = reinterpret_cast<float2 *>(smem + (size_t)(TC + 1) * 32 * 4 * 4) + wave * 64;
    gather_task_body<32, 4, true, OFF32>(tact, kbeg, len, dst, reinterpret_cast<const int64_t *>(work), col, val,
                                         row_map, B, ldb, C, ldc, f, partial, flags, (lane & 31) * 4, mrow);
}

}  // namespace

extern "C" int pgcn_spmm_fused_f32(const int32_t *work, int64_t nwork, const int32_t *col, const float *val,
                                   const int32_t *tasks, const int32_t *row_map, const int32_t *core_work,
                                   const int32_t *tile_panel, const int64_t *tile_base,
                                   const int32_t *seg_off, const int32_t *ccol, const float *cval,
                                   const float *B, int64_t ldb, int64_t ncols, float *C, int64_t ldc,
                                   int32_t f, float *partial_ws, int64_t partial_ws_elems,
                                   int64_t nslots_total, uint32_t flags, pgcn_stream_t stream) {
    if (nwork < 0 || f <= 0 || ldb < f || ldc < f || ncols < 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_fused_f32: bad sizes");
    if (nwork == 0) return PGCN_OK;
    if (f > 128 || f % 4 != 0 || !val || (uintptr_t)B % 16 || (uintptr_t)C % 16 || (uintptr_t)partial_ws % 16 ||
        ldb % 4 || ldc % 4)
        return pgcn_set_error(PGCN_EUNSUPPORTED, "pgcn_spmm_fused_f32: needs f <= 128, f % 4 == 0, values, 16-byte aligned panels");
    if (!work || !col || !tasks || !core_work || !tile_panel || !tile_base || !seg_off || !ccol || !cval || !B || !C ||
        !partial_ws)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_fused_f32: null pointer");
    if (partial_ws_elems < nslots_total * (int64_t)f)
        return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_fused_f32: partial work-space too small");
    if (nwork > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_fused_f32: work list too long");
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = core_smem_bytes(4);
    static bool attr_set = false;
    if (!attr_set) {
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_fused_kernel<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        PGCN_HIP_CHECK(hipFuncSetAttribute((const void *)spmm_fused_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const int4 *w4 = reinterpret_cast<const int4 *>(work);
    const int4 *t4 = reinterpret_cast<const int4 *>(tasks);
    const int4 *c4 = reinterpret_cast<const int4 *>(core_work);
    if (flags & PGCN_SPMM_OFFSETS32)
        hipLaunchKernelGGL((spmm_fused_kernel<true>), dim3((unsigned)nwork), dim3(kCoreThreads), smem, s, w4, col, val,
                           t4, row_map, c4, tile_panel, tile_base, seg_off, ccol, cval, B, ldb, ncols, C, ldc, f,
                           partial_ws, flags);
    else
        hipLaunchKernelGGL((spmm_fused_kernel<false>), dim3((unsigned)nwork), dim3(kCoreThreads), smem, s, w4, col, val,
                           t4, row_map, c4, tile_panel, tile_base, seg_off, ccol, cval, B, ldb, ncols, C, ldc, f,
                           partial_ws, flags);
    PGCN_HIP_CHECK(hipGetLastError());
    return PGCN_OK;
}
