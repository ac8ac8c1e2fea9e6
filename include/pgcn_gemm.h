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

/* ---- the same products as the package's own matrix-core kernels (source: <package>/gemm/pgcn_dense.hip, pgcn_wgrad.hip) ------------
 * v_mfma_f32_32x32x16_bf16 on a three-plane bf16 split of both operands (six partial products, fp32 accumulation: the error
 * class of an fp32 dot product), fused with the element-wise passes either side.  Replaces `F.relu(self.linear(AH))`
 * (GPU/PGCN.py:146-147) and the autograd of those two lines -- all three products since r06.  Row-major operands, fp32, leading
 * dimensions in elements; widths (fin, fout) up to 128; the streamed operands (X; G, Gm) need 16-byte aligned bases, leading
 * dimensions that are multiples of 4 and a width that is a multiple of 4.  Return 0; -2 for operands outside that (nothing was
 * launched: the caller runs the library product); -1 for errors (pgcn_dense_last_error() / pgcn_wgrad_last_error()).  Never
 * allocate, never synchronise.  Binding: <package>/PGCN.py (linear_relu_fused, linear_relu_grad_input_fused, weight_grad_fused),
 * selected by tuning.dense_fused. */
const char *pgcn_dense_last_error(void);

/* Y (n x fout, ldy) = relu ? max(X . W^T, 0) : X . W^T;   X: n x fin (ldx);  W: fout x fin (ldw), nn.Linear's weight.
 * mask (optional): the SIGN MASK of Y, n x ceil(fout / 32) words, bit b of word [row][w] = (Y[row][32 w + b] > 0) -- all the
 * backward needs of Y (r06: 1 bit instead of 4 bytes per element read back). */
int pgcn_linear_relu_f32(const float *X, int64_t ldx, int64_t n, int32_t fin, const float *W, int64_t ldw, int32_t fout,
                         float *Y, int64_t ldy, int32_t relu, uint32_t *mask, void *stream);

/* Gm = G where the mask says Y > 0, else 0 (n x fout, ldgm; written when Gm != NULL -- the operand of the weight gradient; Gm == G is
 * allowed);  dX (n x fin, lddx) = Gm . W.   G: n x fout (ldg);  mask: the forward's sign mask (NULL: Gm = G);  W: fout x fin (ldw) */
int pgcn_linear_relu_grad_input_f32(const float *G, int64_t ldg, const uint32_t *mask, float *Gm, int64_t ldgm, int64_t n,
                                    int32_t fout, const float *W, int64_t ldw, int32_t fin, float *dX, int64_t lddx, void *stream);

/* mask (n x ceil(N / 32) words) = the sign mask of Y (n x N, ldy) in the layout above, for a forward output made elsewhere */
int pgcn_sign_mask_f32(const float *Y, int64_t ldy, int64_t n, int32_t N, uint32_t *mask, void *stream);

/* The third product: dW (fout x fin, lddw) = Gm^T . X, contracted over the n rows;  Gm: n x fout (ldg), X: n x fin (ldx), any
 * alignment.  ws: work-space of at least pgcn_linear_weight_grad_ws_elems() floats (partial matrices per workgroup, added in a fixed
 * order: deterministic, no atomics).  Two launches.  Replaces autograd's `grad_output.t() @ input` of GPU/PGCN.py:146. */
const char *pgcn_wgrad_last_error(void);
int64_t pgcn_linear_weight_grad_ws_elems(void);
int pgcn_linear_weight_grad_f32(const float *Gm, int64_t ldg, const float *X, int64_t ldx, int64_t n, int32_t fout, int32_t fin,
                                float *dW, int64_t lddw, float *ws, int64_t ws_elems, void *stream);

#ifdef __cplusplus
}
#endif
#endif
