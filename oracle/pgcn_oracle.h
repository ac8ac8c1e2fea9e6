/* pgcn_oracle.h -- CPU oracle (TEST INFRASTRUCTURE ONLY, see pgcn_oracle.c). */
#ifndef PGCN_ORACLE_H
#define PGCN_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

void oracle_spmm_csr_f32(int64_t nrows, const int64_t *rowptr, const int32_t *col,
                         const float *val, const float *B, int64_t ldb, float *C,
                         int64_t ldc, int32_t f, int accumulate);
void oracle_spmm_csr_rows_f32(int64_t nsel, const int32_t *rows, const int64_t *rowptr,
                              const int32_t *col, const float *val, const float *B,
                              int64_t ldb, float *C, int64_t ldc, int32_t f, int accumulate);
void oracle_gather_rows_f32(const float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                            float *out, int64_t ldo, int32_t f);
void oracle_scatter_rows_f32(float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                             const float *in, int64_t ldi, int32_t f, int accumulate);
void oracle_dist_aggregate_f32(int64_t n, const int64_t *rowptr, const int32_t *col,
                               const float *val, const int32_t *part, int32_t P,
                               const float *H, int64_t ldh, float *AH, int64_t ldo, int32_t f);
int oracle_pargcn_train(int64_t n, const int64_t *rowptr, const int32_t *col, const float *val,
                        const int32_t *part, int32_t P, int32_t L, const int32_t *d, float **W,
                        const float *H0, const float *Y, const uint8_t *Ymask, int32_t epochs,
                        float alpha, float *err_out, float *Hlast_out, int64_t *stats_out);
int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
