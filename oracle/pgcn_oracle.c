/*
 * pgcn_oracle.c -- CPU restatement of the reference algorithm for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / reported baseline, never as the thing measured or
 * shipped.  The product path (the HIP library) never calls into this file.
 *
 * What is restated (all paths relative to /root/reference):
 *   - Parallel-GCN/main.c:166-454  GCN(): forward 233-316, loss 318-335,
 *     backward 338-438, SGD 430; user ops 66-90.
 *   - the operator both engines share, AH = A_p * H with boundary-row exchange
 *     (Parallel-GCN/main.c:238-299, GPU/PGCN.py:85-127).
 *
 * The reference implements these with SuiteSparse:GraphBLAS (GrB_mxm on the
 * PLUS_TIMES_FP32 semiring etc.).  GraphBLAS is a third-party dependency that
 * is NOT vendored under /root/reference, is not pinned to a version there
 * (Parallel-GCN/Makefile:1-3 just expects ../GraphBLAS) and cannot be fetched
 * (no network).  This file therefore restates the published semantics of the
 * calls the path makes: fp32 storage, fp32 accumulation, PLUS monoid.
 *
 * PARITY PINNING.  The reference ships no tests and no golden outputs.  The
 * aggregation operator is pinned against outputs of the reference's own
 * GPU/PGCN.py (PSpMM forward/backward, compute_communication_maps, full P=1
 * training) generated in the build container by tests/golden/make_golden.py;
 * see tests/test_oracle_golden.py.  The sigmoid/BCE/SGD training loop of
 * main.c is pinned by main.c ITSELF: the file compiles unmodified, from where
 * it lies, against minimal stand-ins for the GraphBLAS / MPI calls it makes
 * (oracle/shim/, `make -C oracle ref` -> oracle/_ref/grbgcn; the stand-ins are
 * restatements of the published interfaces, not SuiteSparse -- what is pinned
 * is main.c's own control flow, operator definitions, message contents and
 * update rule, on top of the textbook meaning of mxm / eWiseAdd / eWiseMult /
 * apply / reduce).  tests/golden/make_pargcn_ref.py runs it on ten data
 * directories (P = 1, 2, 3, 4; 2, 3 and 4 layers; three of them written by the
 * reference's own preprocess + GCN-HP tools) and commits what it printed and
 * the weights it ended with; oracle_pargcn_train ends on the same weights BIT
 * FOR BIT (tests/test_reference_grbgcn.py).  That run also showed a property
 * of the reference this file now follows: on an unsymmetric pattern the send
 * lists GCN-HP writes are not what the receivers' rows need, and main.c
 * ignores the entries whose rows never arrive (oracle.drop_undelivered).
 *
 * Summation order.  GraphBLAS leaves the order of the PLUS reduction to its
 * kernels and main.c:278 accumulates remote pieces in message-arrival order,
 * so the reference itself is only defined up to fp32 re-association.  This
 * restatement fixes ONE valid order: within a row, stored (CSR) order; local
 * piece first (main.c:271), then remote pieces by ascending source rank
 * (main.c:295).  Comparisons against it use a relative tolerance of 1e-5.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "pgcn_oracle.h"

/* ------------------------------------------------------------------ */
/* user-defined ops, Parallel-GCN/main.c:66-90                          */

/* main.c:79-81 sigmoid(): 1 / (1 + expf(-x)) in float */
static inline float op_sigmoid(float x) { return 1 / (1 + expf(-x)); }

/* main.c:83-86 sigmoid_derivative(): s = sigmoid(x); s * (1 - s) */
static inline float op_sigmoid_derivative(float x) {
    float s = op_sigmoid(x);
    return s * (1 - s);
}

/* main.c:88-90 cross_entropy_derivative_divisor(): x * (1 - x) */
static inline float op_ced_divisor(float x) { return x * (1 - x); }

/* main.c:70-73 binary_cross_entropy_loss(): -1 * y * log(x); `log` is the
 * double-precision libm log applied to a float promoted to double, the
 * product is formed in double and rounded to float on store. */
static inline float op_bce(float x, float y) { return (float)(-1 * y * log(x)); }

/* main.c:75-77 gradient_update(): x - alpha * y */
static inline float op_gradient_update(float x, float y, float alpha) { return x - alpha * y; }

/* ------------------------------------------------------------------ */
/* CSR SpMM: C (+)= A * B, fp32, PLUS_TIMES_FP32 (main.c:271, 295, 376, 400) */

void oracle_spmm_csr_f32(int64_t nrows, const int64_t *rowptr, const int32_t *col,
                         const float *val, const float *B, int64_t ldb, float *C,
                         int64_t ldc, int32_t f, int accumulate) {
    int64_t i;
#pragma omp parallel for schedule(dynamic, 64)
    for (i = 0; i < nrows; i++) {
        float *c = C + i * ldc;
        if (!accumulate)
            for (int32_t j = 0; j < f; j++) c[j] = 0.0f;
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
            const float a = val[k];
            const float *b = B + (int64_t)col[k] * ldb;
            for (int32_t j = 0; j < f; j++) c[j] += a * b[j];
        }
    }
}

/* Row-subset variant: only rows listed in `rows` are (re)computed. */
void oracle_spmm_csr_rows_f32(int64_t nsel, const int32_t *rows, const int64_t *rowptr,
                              const int32_t *col, const float *val, const float *B,
                              int64_t ldb, float *C, int64_t ldc, int32_t f,
                              int accumulate) {
    int64_t r;
#pragma omp parallel for schedule(dynamic, 64)
    for (r = 0; r < nsel; r++) {
        const int64_t i = rows[r];
        float *c = C + i * ldc;
        if (!accumulate)
            for (int32_t j = 0; j < f; j++) c[j] = 0.0f;
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
            const float a = val[k];
            const float *b = B + (int64_t)col[k] * ldb;
            for (int32_t j = 0; j < f; j++) c[j] += a * b[j];
        }
    }
}

/* pack / unpack of boundary rows.
 * gather  = GPU/PGCN.py:104  H[indices]            (== the row select
 *           Hsend[q] (x) H of main.c:250 followed by extractTuples 254)
 * scatter = GPU/PGCN.py:115  X[indices] = buf  (accumulate=0)
 *           or the accumulate-on-receive of main.c:295,400 (accumulate=1). */
void oracle_gather_rows_f32(const float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                            float *out, int64_t ldo, int32_t f) {
    for (int64_t r = 0; r < nidx; r++)
        memcpy(out + r * ldo, H + (int64_t)idx[r] * ldh, (size_t)f * sizeof(float));
}

void oracle_scatter_rows_f32(float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                             const float *in, int64_t ldi, int32_t f, int accumulate) {
    for (int64_t r = 0; r < nidx; r++) {
        float *h = H + (int64_t)idx[r] * ldh;
        const float *s = in + r * ldi;
        if (accumulate)
            for (int32_t j = 0; j < f; j++) h[j] += s[j];
        else
            for (int32_t j = 0; j < f; j++) h[j] = s[j];
    }
}

/* ------------------------------------------------------------------ */
/* distributed aggregation with P virtual ranks (main.c:238-299):
 *   rank p = part[i] owns row i.  AH[i,:] = sum over the LOCAL columns
 *   (main.c:271: H holds only owned rows, so A (x) H picks up exactly the
 *   columns owned by p), then += one separately-formed product per source
 *   rank q (main.c:293-295: Hcap = received rows of q; AH += A (x) Hcap),
 *   sources taken in ascending rank order. */
void oracle_dist_aggregate_f32(int64_t n, const int64_t *rowptr, const int32_t *col,
                               const float *val, const int32_t *part, int32_t P,
                               const float *H, int64_t ldh, float *AH, int64_t ldo,
                               int32_t f) {
    int64_t i;
#pragma omp parallel
    {
        float *t = (float *)malloc((size_t)f * sizeof(float));
#pragma omp for schedule(dynamic, 64)
        for (i = 0; i < n; i++) {
            const int32_t p = part[i];
            float *c = AH + i * ldo;
            for (int32_t j = 0; j < f; j++) c[j] = 0.0f;
            /* local piece, main.c:271 */
            for (int64_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
                if (part[col[k]] != p) continue;
                const float a = val[k];
                const float *b = H + (int64_t)col[k] * ldh;
                for (int32_t j = 0; j < f; j++) c[j] += a * b[j];
            }
            /* remote pieces, main.c:275-299, one product per source then accum */
            for (int32_t q = 0; q < P; q++) {
                if (q == p) continue;
                int any = 0;
                for (int64_t k = rowptr[i]; k < rowptr[i + 1]; k++) {
                    if (part[col[k]] != q) continue;
                    if (!any) {
                        for (int32_t j = 0; j < f; j++) t[j] = 0.0f;
                        any = 1;
                    }
                    const float a = val[k];
                    const float *b = H + (int64_t)col[k] * ldh;
                    for (int32_t j = 0; j < f; j++) t[j] += a * b[j];
                }
                if (any)
                    for (int32_t j = 0; j < f; j++) c[j] += t[j];
            }
        }
        free(t);
    }
}

/* dense helpers, row-major, fp32 accumulate in k order */
/* C[m x n] = A[m x k] * B[k x n]          (main.c:303  Z = AH (x) W) */
static void gemm_nn(int64_t m, int32_t k, int32_t n, const float *A, const float *B, float *C) {
    int64_t i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < m; i++) {
        float *c = C + i * n;
        for (int32_t j = 0; j < n; j++) c[j] = 0.0f;
        for (int32_t l = 0; l < k; l++) {
            const float a = A[i * k + l];
            const float *b = B + (int64_t)l * n;
            for (int32_t j = 0; j < n; j++) c[j] += a * b[j];
        }
    }
}

/* C[m x k] = A[m x n] * B[k x n]^T        (main.c:407  G = AG (x) W^T, DESC_RT1) */
static void gemm_nt(int64_t m, int32_t n, int32_t k, const float *A, const float *B, float *C) {
    int64_t i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < m; i++) {
        for (int32_t l = 0; l < k; l++) {
            float s = 0.0f;
            for (int32_t j = 0; j < n; j++) s += A[i * n + j] * B[(int64_t)l * n + j];
            C[i * k + l] = s;
        }
    }
}

/* ------------------------------------------------------------------ */
/* Full training loop of Parallel-GCN/main.c:GCN() with P virtual ranks.
 *
 *  d[0..L]       nneurons (main.c:692-698): d[0] = n, number of GCN layers = L-1
 *  W[l]          l = 1..L-1, d[l] x d[l+1] row-major, updated in place (main.c:430)
 *  H0            n x d[1] input features            (main.c:650-685: ones)
 *  Y, Ymask      n x d[L]; Ymask[i,j] != 0 where the sparse Y holds an entry
 *                (preprocess/GrB-GNN-IDG.py:76-78 stores only column 1)
 *  err_out[e]    the value rank 0 prints as "err:%g" per epoch (main.c:318-323)
 *  Hlast_out     optional n x d[L]: H_{L-1} (the network output) of the LAST epoch
 *  stats_out     optional [2*P]: per rank send volume (scalars) and message count
 */
int oracle_pargcn_train(int64_t n, const int64_t *rowptr, const int32_t *col,
                        const float *val, const int32_t *part, int32_t P, int32_t L,
                        const int32_t *d, float **W, const float *H0, const float *Y,
                        const uint8_t *Ymask, int32_t epochs, float alpha, float *err_out,
                        float *Hlast_out, int64_t *stats_out) {
    if (L < 2 || d[0] != n) return -1;
    int32_t fmax = 0;
    for (int32_t l = 1; l <= L; l++)
        if (d[l] > fmax) fmax = d[l];

    /* H[l], Z[l] for l = 0..L-1 with widths d[l+1] (main.c:567-569, 301) */
    float **H = (float **)calloc((size_t)L, sizeof(float *));
    float **Z = (float **)calloc((size_t)L, sizeof(float *));
    float **G = (float **)calloc((size_t)L, sizeof(float *));
    for (int32_t l = 0; l < L; l++) {
        H[l] = (float *)malloc((size_t)n * d[l + 1] * sizeof(float));
        Z[l] = (float *)malloc((size_t)n * d[l + 1] * sizeof(float));
        G[l] = (float *)malloc((size_t)n * d[l + 1] * sizeof(float));
    }
    float *AH = (float *)malloc((size_t)n * fmax * sizeof(float));
    float *T = (float *)malloc((size_t)n * d[L] * sizeof(float));
    float *dWp = (float *)malloc((size_t)fmax * fmax * sizeof(float));
    float *dW = (float *)malloc((size_t)fmax * fmax * sizeof(float));
    memcpy(H[0], H0, (size_t)n * d[1] * sizeof(float));

    /* message statistics: rows p->q = #cols owned by p referenced by rows of q */
    int64_t *rows_pq = (int64_t *)calloc((size_t)P * P, sizeof(int64_t));
    {
        /* mark[j*P + q] = column j is needed by rank q */
        uint8_t *mark = (uint8_t *)calloc((size_t)n * P, 1);
        for (int64_t i = 0; i < n; i++)
            for (int64_t k = rowptr[i]; k < rowptr[i + 1]; k++)
                if (part[col[k]] != part[i]) mark[(int64_t)col[k] * P + part[i]] = 1;
        for (int64_t j = 0; j < n; j++)
            for (int32_t q = 0; q < P; q++)
                if (mark[j * P + q]) rows_pq[(int64_t)part[j] * P + q]++;
        free(mark);
    }
    if (stats_out) memset(stats_out, 0, (size_t)2 * P * sizeof(int64_t));

    for (int32_t epoch = 0; epoch < epochs; epoch++) { /* main.c:231 */
        /* ---------------- forward, main.c:233-316 ---------------- */
        for (int32_t layer = 1; layer < L; layer++) {
            const int32_t fi = d[layer], fo = d[layer + 1];
            oracle_dist_aggregate_f32(n, rowptr, col, val, part, P, H[layer - 1], fi, AH, fi, fi);
            if (stats_out)
                for (int32_t p = 0; p < P; p++)
                    for (int32_t q = 0; q < P; q++)
                        if (p != q && rows_pq[(int64_t)p * P + q]) {
                            stats_out[2 * p] += rows_pq[(int64_t)p * P + q] * fi; /* main.c:264 nvals */
                            stats_out[2 * p + 1] += 1;                           /* main.c:265 */
                        }
            gemm_nn(n, fi, fo, AH, W[layer], Z[layer]); /* main.c:303 */
            for (int64_t e = 0; e < n * (int64_t)fo; e++) H[layer][e] = op_sigmoid(Z[layer][e]); /* 308 */
        }
        /* ---------------- loss, main.c:318-323 ---------------- */
        const int32_t fl = d[L];
        float *Pm = H[L - 1];
        float t_err = 0.0f;
        for (int32_t p = 0; p < P; p++) { /* per-rank GrB_reduce, then MPI_Reduce SUM */
            float err = 0.0f;
            for (int64_t i = 0; i < n; i++) {
                if (part[i] != p) continue;
                for (int32_t j = 0; j < fl; j++) {
                    const int64_t e = i * fl + j;
                    /* eWiseAdd = set UNION: both present -> op; only H present -> H */
                    T[e] = Ymask[e] ? op_bce(Pm[e], Y[e]) : Pm[e];
                    err += T[e];
                }
            }
            t_err += err;
        }
        err_out[epoch] = t_err;
        if (Hlast_out && epoch == epochs - 1) memcpy(Hlast_out, Pm, (size_t)n * fl * sizeof(float));
        /* ---------------- output gradient, main.c:325-335 ---------------- */
        for (int64_t e = 0; e < n * (int64_t)fl; e++) {
            const float t = op_ced_divisor(Pm[e]);              /* 325 */
            float h = Ymask[e] ? Pm[e] - Y[e] : Pm[e];          /* 327 eWiseAdd MINUS (union) */
            h = h / t;                                          /* 328 */
            const float zp = op_sigmoid_derivative(Z[L - 1][e]); /* 330 */
            G[L - 1][e] = (h * zp) / (float)d[0];               /* 331, 335 */
        }
        /* ---------------- backward, main.c:338-438 ---------------- */
        for (int32_t layer = L - 1; layer > 0; layer--) {
            const int32_t fi = d[layer], fo = d[layer + 1];
            /* AG = A (x) G[layer] with the same exchange (343-404); uses A, not A^T */
            oracle_dist_aggregate_f32(n, rowptr, col, val, part, P, G[layer], fo, AH, fo, fo);
            if (stats_out)
                for (int32_t p = 0; p < P; p++)
                    for (int32_t q = 0; q < P; q++)
                        if (p != q && rows_pq[(int64_t)p * P + q]) {
                            stats_out[2 * p] += rows_pq[(int64_t)p * P + q] * fo;
                            stats_out[2 * p + 1] += 1;
                        }
            if (layer != 1) { /* main.c:406-411 */
                gemm_nt(n, fo, fi, AH, W[layer], G[layer - 1]);
                for (int64_t e = 0; e < n * (int64_t)fi; e++)
                    G[layer - 1][e] = G[layer - 1][e] * op_sigmoid_derivative(Z[layer - 1][e]);
            }
            /* dW = H[layer-1]^T (x) AG per rank (415-418), Allreduce SUM (425) */
            memset(dW, 0, (size_t)fi * fo * sizeof(float));
            for (int32_t p = 0; p < P; p++) {
                memset(dWp, 0, (size_t)fi * fo * sizeof(float));
                /* every thread owns rows `a` of dW and walks the vertices in ascending order: the additions into one
                 * element happen in the order of the serial loop (bit-identical to it), all host cores busy */
#pragma omp parallel for schedule(static)
                for (int32_t a = 0; a < fi; a++) {
                    float *dst = dWp + (int64_t)a * fo;
                    for (int64_t i = 0; i < n; i++) {
                        if (part[i] != p) continue;
                        const float ha = H[layer - 1][i * fi + a];
                        const float *g = AH + i * fo;
                        for (int32_t b = 0; b < fo; b++) dst[b] += ha * g[b];
                    }
                }
                for (int32_t e = 0; e < fi * fo; e++) dW[e] += dWp[e];
            }
            for (int32_t e = 0; e < fi * fo; e++) /* main.c:430 */
                W[layer][e] = op_gradient_update(W[layer][e], dW[e], alpha);
        }
    }

    for (int32_t l = 0; l < L; l++) {
        free(H[l]);
        free(Z[l]);
        free(G[l]);
    }
    free(H); free(Z); free(G); free(AH); free(T); free(dWp); free(dW); free(rows_pq);
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
