/* TEST INFRASTRUCTURE -- not product code, never linked into the HIP library.
 *
 * Minimal stand-in for the part of the GraphBLAS C API (v1.3, with the SuiteSparse GxB_ names) that the
 * reference's CPU engine /root/reference/Parallel-GCN/main.c calls.  SuiteSparse:GraphBLAS is a
 * dependency of the reference that is neither vendored under /root/reference nor installed here, so the
 * reference cannot be built as its Makefile says.  With this header (and oracle/shim/mpi.h) on the
 * include path the reference's OWN, UNMODIFIED main.c compiles where it lies (oracle/Makefile, target
 * _ref/grbgcn); the binary's output pins oracle/pgcn_oracle.c::oracle_pargcn_train
 * (tests/golden/make_pargcn_ref.py -> tests/golden/pargcn_ref_*.{json,npz}).
 *
 * What this is and is not: every function below follows the mathematical definition in the GraphBLAS
 * C API specification (matrices as sets of (i, j, value) tuples; mxm = T over a semiring, then the
 * accumulate / mask / replace write-back of section 2.4 -- masks are always NULL in main.c);
 * floating-point sums run in ascending k / ascending (i, j) order, which the specification leaves open
 * (SuiteSparse's own order depends on its method and thread count), so the last bits of a sum may differ
 * from a SuiteSparse build.  Only FP32 matrices exist.  It is a restatement of a published interface,
 * written from scratch; nothing here is copied from SuiteSparse.
 *
 * One deliberate note: main.c:218 creates its PLUS monoid with identity `true` (= 1.0f).  The
 * specification requires the identity of the operator; behaviour with another value is undefined
 * (SuiteSparse would add it once per reduction task).  The shim never uses the identity of a monoid for
 * a non-empty reduction, i.e. `err` is the plain sum.
 */
#ifndef PGCN_ORACLE_SHIM_GRAPHBLAS_H
#define PGCN_ORACLE_SHIM_GRAPHBLAS_H

/* the real GraphBLAS.h pulls these in; main.c relies on that for strcpy / strlen */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef uint64_t GrB_Index;

typedef enum {
    GrB_SUCCESS = 0, GrB_NO_VALUE = 1, GrB_UNINITIALIZED_OBJECT = 2, GrB_INVALID_OBJECT = 3, GrB_NULL_POINTER = 4,
    GrB_INVALID_VALUE = 5, GrB_INVALID_INDEX = 6, GrB_DOMAIN_MISMATCH = 7, GrB_DIMENSION_MISMATCH = 8,
    GrB_OUTPUT_NOT_EMPTY = 9, GrB_OUT_OF_MEMORY = 10, GrB_INSUFFICIENT_SPACE = 11, GrB_INDEX_OUT_OF_BOUNDS = 12,
    GrB_PANIC = 13
} GrB_Info;

typedef enum { GrB_NONBLOCKING = 0, GrB_BLOCKING = 1 } GrB_Mode;
typedef enum { GxB_NTHREADS = 5 } GxB_Option_Field;

typedef struct grbshim_type *GrB_Type;
typedef struct grbshim_matrix *GrB_Matrix;
typedef struct grbshim_unop *GrB_UnaryOp;
typedef struct grbshim_binop *GrB_BinaryOp;
typedef struct grbshim_monoid *GrB_Monoid;
typedef struct grbshim_semiring *GrB_Semiring;
typedef struct grbshim_desc *GrB_Descriptor;

typedef void (*GxB_unary_function)(void *, const void *);
typedef void (*GxB_binary_function)(void *, const void *, const void *);

#define GrB_NULL NULL
extern const GrB_Index *GrB_ALL;
extern GrB_Type GrB_FP32;
extern GrB_BinaryOp GrB_PLUS_FP32, GrB_MINUS_FP32, GrB_TIMES_FP32, GrB_DIV_FP32;
extern GrB_Semiring GxB_PLUS_TIMES_FP32, GxB_PLUS_SECOND_FP32;
extern GrB_Descriptor GrB_DESC_R, GrB_DESC_T0, GrB_DESC_RT1;

GrB_Info GrB_init(GrB_Mode mode);
GrB_Info GxB_init(GrB_Mode mode, void *(*user_malloc)(size_t), void *(*user_calloc)(size_t, size_t),
                  void *(*user_realloc)(void *, size_t), void (*user_free)(void *), bool thread_safe);
GrB_Info GxB_set(GxB_Option_Field field, ...);
GrB_Info GrB_finalize(void);

GrB_Info GrB_UnaryOp_new(GrB_UnaryOp *op, GxB_unary_function fn, GrB_Type ztype, GrB_Type xtype);
GrB_Info GrB_BinaryOp_new(GrB_BinaryOp *op, GxB_binary_function fn, GrB_Type ztype, GrB_Type xtype, GrB_Type ytype);
GrB_Info GrB_Monoid_new_FP32(GrB_Monoid *monoid, GrB_BinaryOp op, float identity);

GrB_Info GrB_Matrix_new(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols);
GrB_Info GrB_Matrix_free(GrB_Matrix *A);
GrB_Info GrB_Matrix_clear(GrB_Matrix A);
GrB_Info GrB_Matrix_wait(GrB_Matrix *A);
GrB_Info GrB_Matrix_nvals(GrB_Index *nvals, const GrB_Matrix A);
GrB_Info GrB_Matrix_build_FP32(GrB_Matrix C, const GrB_Index *I, const GrB_Index *J, const float *X, GrB_Index nvals,
                               const GrB_BinaryOp dup);
GrB_Info GrB_Matrix_setElement_FP64(GrB_Matrix C, double x, GrB_Index i, GrB_Index j);
GrB_Info GrB_Matrix_extractTuples_FP32(GrB_Index *I, GrB_Index *J, float *X, GrB_Index *nvals, const GrB_Matrix A);

GrB_Info GrB_mxm(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                      const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseMult_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                       const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_apply(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_UnaryOp op,
                          const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_apply_BinaryOp2nd_FP32(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                           const GrB_Matrix A, float y, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_assign_FP32(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, float x, const GrB_Index *I,
                                GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_reduce_FP32(float *c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A,
                                const GrB_Descriptor desc);

/* the polymorphic names of the specification, resolved for the only forms main.c uses (FP32 matrices, BinaryOp operators) */
#define GrB_Matrix_build GrB_Matrix_build_FP32
#define GrB_Matrix_setElement(C, x, i, j) GrB_Matrix_setElement_FP64(C, (double) (x), i, j)
#define GrB_Matrix_extractTuples GrB_Matrix_extractTuples_FP32
#define GrB_eWiseAdd GrB_Matrix_eWiseAdd_BinaryOp
#define GrB_eWiseMult GrB_Matrix_eWiseMult_BinaryOp
#define GrB_apply GrB_Matrix_apply
#define GrB_reduce GrB_Matrix_reduce_FP32

#endif
