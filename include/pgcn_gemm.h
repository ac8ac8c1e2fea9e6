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

/* rocBLAS atomics mode of the handles behind pgcn_gemm_f32 (one handle per (device, stream): concurrent streams do not share a
 * device work-space): 1 = allowed (default, PyTorch's default), 0 = not allowed (what torch.use_deterministic_algorithms(True)
 * sets on PyTorch's own handles; the binding copies that flag before every product). */
void pgcn_gemm_set_atomics(int32_t allowed);

/* ---- the same products as the package's own matrix-core kernels (source: <package>/gemm/pgcn_dense.hip) -----------------------
 * v_mfma_f32_32x32x16_bf16 on a three-plane bf16 split of both operands (six partial products, fp32 accumulation: the error
 * class of an fp32 dot product), one persistent workgroup per CU with the split weight matrix in LDS, fused with the
 * element-wise passes either side.  Replaces `F.relu(self.linear(AH))` (GPU/PGCN.py:146-147) and the autograd of those two
 * lines.  Row-major operands, fp32, leading dimensions in elements; widths (fin, fout) up to 128; the streamed operands
 * (X; G, Y, Gm) need 16-byte aligned bases, leading dimensions that are multiples of 4 and a width that is a multiple of 4.
 * Return 0; -2 for operands outside that (nothing was launched: the caller runs the library product); -1 for errors
 * (pgcn_dense_last_error()).  Binding: <package>/PGCN.py (linear_relu_fused, linear_relu_grad_input_fused), selected by
 * tuning.dense_fused. */
const char *pgcn_dense_last_error(void);

/* Y (n x fout, ldy) = relu ? max(X . W^T, 0) : X . W^T;   X: n x fin (ldx);  W: fout x fin (ldw), nn.Linear's weight */
int pgcn_linear_relu_f32(const float *X, int64_t ldx, int64_t n, int32_t fin, const float *W, int64_t ldw, int32_t fout,
                         float *Y, int64_t ldy, int32_t relu, void *stream);

/* Gm = G where Y > 0, else 0 (n x fout, ldgm; written when Gm != NULL -- the operand of the weight gradient; Gm == G is
 * allowed);  dX (n x fin, lddx) = Gm . W.   G, Y: n x fout (ldg, ldy);  W: fout x fin (ldw) */
int pgcn_linear_relu_grad_input_f32(const float *G, int64_t ldg, const float *Y, int64_t ldy, float *Gm, int64_t ldgm,
                                    int64_t n, int32_t fout, const float *W, int64_t ldw, int32_t fin, float *dX,
                                    int64_t lddx, void *stream);

/* C (n x N, ldc) = epi(X . Bm):  X: n x k (ldx);  W: wrows x wcols (ldw);  transposed 1: Bm = W^T (N = wrows, wcols = k), 0: Bm = W
 * (wrows = k, N = wcols);  epilogue 0: none, 1: relu, 2: keep the product where M (n x N, ldm) > 0, else 0 -- a layer's input
 * gradient with the ReLU mask of the layer below folded in (threshold_backward by that layer's output). */
int pgcn_linear_epilogue_f32(const float *X, int64_t ldx, int64_t n, int32_t k, const float *W, int64_t ldw, int32_t wrows,
                             int32_t wcols, int32_t transposed, const float *M, int64_t ldm, float *C, int64_t ldc,
                             int32_t epilogue, void *stream);

/* The same product with the FIX-UP of the aggregation as its loader (r05): the left operand is never materialised.
 *   S[r] = count >= 0 ? ((0 + P[id_0]) + P[id_1]) + ... : base[r],   row_fix[r] = {begin, count} (n x 2 int32),
 *   id_t = slot_ids[begin + t] (begin + t when slot_ids == NULL),  P[i] = partial + i * ldp  (the producers' work-space:
 *   include/pgcn_hip.h, pgcn_spmm_*_f32 with PGCN_SPMM_NO_FIXUP), summed in list order exactly like pgcn_spmm_fixup_f32 does;
 *   C = epi(S . Bm) as above;  S_out (n x k, lds) != NULL: S is also written (the operand of the weight gradient).
 * Replaces pgcn_spmm_fixup_f32 + the product on its output: `H = PSpMM.apply(A, H); F.relu(self.linear(H))`,
 * /root/reference/GPU/PGCN.py:144-147, and the same pair in the backward.  Bit-identical to that pair.  k a multiple of 4, widths
 * up to 128, partial / base / S_out rows 16-byte pieces; -2 otherwise (nothing launched). */
int pgcn_fixup_linear_f32(const int32_t *row_fix, const int32_t *slot_ids, const float *partial, int64_t ldp, const float *base,
                          int64_t ldbase, int64_t n, int32_t k, const float *W, int64_t ldw, int32_t wrows, int32_t wcols,
                          int32_t transposed, float *S_out, int64_t lds, const float *M, int64_t ldm, float *C, int64_t ldc,
                          int32_t epilogue, void *stream);

/* The third product of the layer, dW = Gm^T . X, stays the library's 64-slab batched GEMM (78 us at n = 232 965, f = 128); the
 * package's own kernel for it measured 500 + 58 us on the MI355X in r05 and was moved out of the library
 * (tools/experiments/pgcn_wgrad.hip). */

#ifdef __cplusplus
}
#endif
#endif
