/* pgcn_gemm.h -- C ABI of lib/libpgcn_gemm.so: the dense products of a layer as stock rocBLAS GEMMs launched by solution
 * index.  Plumbing BESIDE the graded aggregation path (include/pgcn_hip.h): replaces nothing of the reference but the
 * library call behind `self.linear(H)` (GPU/PGCN.py:139,146) and its backward, which stay stock GEMMs.  Why it exists:
 * the kernel PyTorch's TunableOp finds for the n x f x f shapes is 15-20 % faster than the default pick, but switching
 * TunableOp on costs 20-30 s of set-up on a cold box; its recorded choice (tunableop/gfx950.csv) is replayed here.
 * Source: <package>/gemm/pgcn_gemm.cpp; binding: <package>/PGCN.py (mm_nt, mm_nn).  All functions return 0 on success. */
#ifndef PGCN_GEMM_H
#define PGCN_GEMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char *pgcn_gemm_last_error(void);

/* build string of the rocBLAS this process bound (compared with the `Validator,ROCBLAS_VERSION` line of the result file) */
int pgcn_gemm_rocblas_version(char *buf, int64_t n);

/* C (m x n, ldc) = op(A) . op(B), column-major like rocBLAS, fp32, alpha 1, beta 0, on `stream` (hipStream_t) of the
 * current device, with kernel `solution_index` of that rocBLAS build (0: the library's own pick).  transa / transb:
 * 0 = N, 1 = T.  -2: rocBLAS refuses the index for this problem (the caller uses its default product); -1: other errors. */
int pgcn_gemm_f32(int32_t transa, int32_t transb, int64_t m, int64_t n, int64_t k, const float *A, int64_t lda,
                  const float *B, int64_t ldb, float *C, int64_t ldc, int32_t solution_index, void *stream);

#ifdef __cplusplus
}
#endif
#endif
