/* TEST INFRASTRUCTURE -- see oracle/shim/GraphBLAS.h for what this is and why it exists.
 *
 * FP32 matrices as CSR (rows ascending, columns ascending inside a row) plus a list of pending
 * setElement tuples that is merged on first use.  Every operation computes its result T into a fresh
 * CSR and then writes it back as the specification's section 2.4 says for a NULL mask:
 *     accum == NULL :  C = T                      (REPLACE has no effect without a mask)
 *     accum != NULL :  C = C (+) T   over the UNION of the two patterns (accum where both exist)
 * Sums: mxm adds the products of one output entry in ascending k; reduce folds in (i, j) order.
 *
 * GRBSHIM_DUMP=<prefix> in the environment: GrB_finalize writes every matrix that is still alive, in the
 * order of creation, to <prefix>.<MPISHIM_RANK> (binary: "GRBD", count, then per matrix
 * serial / nrows / ncols / nvals as int64 followed by I[], J[] (int64) and X[] (float)) -- that is how
 * the final weights W[l] of a run of the reference leave the process (main.c prints only err / time / stats). */
#include "GraphBLAS.h"
#include <stdarg.h>

struct grbshim_type { int code; };
struct grbshim_unop { GxB_unary_function fn; };
struct grbshim_binop { GxB_binary_function fn; };
struct grbshim_monoid { GrB_BinaryOp op; float identity; };
struct grbshim_semiring { int second; };           /* add = PLUS; multiply = TIMES (0) or SECOND (1) */
struct grbshim_desc { int replace, t0, t1; };

typedef struct { int64_t *p, *j; float *x; int64_t nvals; } csr_t;

struct grbshim_matrix {
    int64_t nrows, ncols;
    csr_t c;
    int64_t npend, cappend, *pi, *pj;
    float *px;
    int64_t serial;
    struct grbshim_matrix *prev, *next;
};

static struct grbshim_type t_fp32 = {1};
GrB_Type GrB_FP32 = &t_fp32;
static const GrB_Index all_marker = 0;
const GrB_Index *GrB_ALL = &all_marker;

static void f_plus(void *z, const void *x, const void *y) { *(float *) z = *(const float *) x + *(const float *) y; }
static void f_minus(void *z, const void *x, const void *y) { *(float *) z = *(const float *) x - *(const float *) y; }
static void f_times(void *z, const void *x, const void *y) { *(float *) z = *(const float *) x * *(const float *) y; }
static void f_div(void *z, const void *x, const void *y) { *(float *) z = *(const float *) x / *(const float *) y; }
static struct grbshim_binop b_plus = {f_plus}, b_minus = {f_minus}, b_times = {f_times}, b_div = {f_div};
GrB_BinaryOp GrB_PLUS_FP32 = &b_plus, GrB_MINUS_FP32 = &b_minus, GrB_TIMES_FP32 = &b_times, GrB_DIV_FP32 = &b_div;
static struct grbshim_semiring s_pt = {0}, s_ps = {1};
GrB_Semiring GxB_PLUS_TIMES_FP32 = &s_pt, GxB_PLUS_SECOND_FP32 = &s_ps;
static struct grbshim_desc d_r = {1, 0, 0}, d_t0 = {0, 1, 0}, d_rt1 = {1, 0, 1};
GrB_Descriptor GrB_DESC_R = &d_r, GrB_DESC_T0 = &d_t0, GrB_DESC_RT1 = &d_rt1;

static struct grbshim_matrix *live_head = NULL, *live_tail = NULL;
static int64_t next_serial = 0;

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "grb_shim: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "grb_shim: out of memory (%zu bytes)\n", n); abort(); }
    return p;
}

static csr_t csr_empty(int64_t nrows) {
    csr_t c;
    c.p = (int64_t *) xmalloc((size_t) (nrows + 1) * sizeof(int64_t));
    memset(c.p, 0, (size_t) (nrows + 1) * sizeof(int64_t));
    c.j = NULL; c.x = NULL; c.nvals = 0;
    return c;
}
static void csr_free(csr_t *c) { free(c->p); free(c->j); free(c->x); c->p = c->j = NULL; c->x = NULL; c->nvals = 0; }

/* output CSR under construction */
typedef struct { csr_t c; int64_t cap; } out_t;
static out_t out_begin(int64_t nrows, int64_t guess) {
    out_t o;
    o.c.p = (int64_t *) xmalloc((size_t) (nrows + 1) * sizeof(int64_t));
    o.c.p[0] = 0;
    o.cap = guess > 16 ? guess : 16;
    o.c.j = (int64_t *) xmalloc((size_t) o.cap * sizeof(int64_t));
    o.c.x = (float *) xmalloc((size_t) o.cap * sizeof(float));
    o.c.nvals = 0;
    return o;
}
static inline void out_push(out_t *o, int64_t j, float x) {
    if (o->c.nvals == o->cap) {
        o->cap *= 2;
        o->c.j = (int64_t *) xrealloc(o->c.j, (size_t) o->cap * sizeof(int64_t));
        o->c.x = (float *) xrealloc(o->c.x, (size_t) o->cap * sizeof(float));
    }
    o->c.j[o->c.nvals] = j; o->c.x[o->c.nvals] = x; o->c.nvals++;
}

/* ---- tuples -> CSR --------------------------------------------------------------------------------- */
static const int64_t *srt_i, *srt_j;
static int cmp_tuple(const void *a, const void *b) {
    const int64_t u = *(const int64_t *) a, v = *(const int64_t *) b;
    if (srt_i[u] != srt_i[v]) return srt_i[u] < srt_i[v] ? -1 : 1;
    if (srt_j[u] != srt_j[v]) return srt_j[u] < srt_j[v] ? -1 : 1;
    return u < v ? -1 : (u > v);                      /* input order among duplicates */
}

/* dup == NULL: a later tuple overwrites an earlier one (setElement); else duplicates are folded left to right. */
static GrB_Info csr_from_tuples(csr_t *res, int64_t nrows, int64_t ncols, const int64_t *I, const int64_t *J, const float *X,
                                int64_t n, GxB_binary_function dup) {
    int sorted = 1;
    for (int64_t k = 0; k < n; ++k) {
        if (I[k] < 0 || I[k] >= nrows || J[k] < 0 || J[k] >= ncols) return GrB_INDEX_OUT_OF_BOUNDS;
        if (k && (I[k - 1] > I[k] || (I[k - 1] == I[k] && J[k - 1] > J[k]))) sorted = 0;
    }
    int64_t *ord = NULL;
    if (!sorted) {
        ord = (int64_t *) xmalloc((size_t) n * sizeof(int64_t));
        for (int64_t k = 0; k < n; ++k) ord[k] = k;
        srt_i = I; srt_j = J;
        qsort(ord, (size_t) n, sizeof(int64_t), cmp_tuple);
    }
    out_t o = out_begin(nrows, n);
    int64_t row = 0, li = -1, lj = -1;
    for (int64_t q = 0; q < n; ++q) {
        const int64_t k = ord ? ord[q] : q;
        if (I[k] == li && J[k] == lj) {
            float *z = &o.c.x[o.c.nvals - 1];
            if (dup) { float t; dup(&t, z, &X[k]); *z = t; } else *z = X[k];
            continue;
        }
        while (row < I[k]) o.c.p[++row] = o.c.nvals;
        out_push(&o, J[k], X[k]);
        li = I[k]; lj = J[k];
    }
    while (row < nrows) o.c.p[++row] = o.c.nvals;
    free(ord);
    *res = o.c;
    return GrB_SUCCESS;
}

/* C (+) T over the union of the patterns; op == NULL keeps T where both exist (used to merge pending tuples). */
static csr_t csr_union(int64_t nrows, const csr_t *a, const csr_t *b, GxB_binary_function op) {
    out_t o = out_begin(nrows, a->nvals + b->nvals);
    for (int64_t i = 0; i < nrows; ++i) {
        int64_t pa = a->p[i], pb = b->p[i];
        const int64_t ea = a->p[i + 1], eb = b->p[i + 1];
        while (pa < ea || pb < eb) {
            if (pb >= eb || (pa < ea && a->j[pa] < b->j[pb])) { out_push(&o, a->j[pa], a->x[pa]); ++pa; }
            else if (pa >= ea || b->j[pb] < a->j[pa]) { out_push(&o, b->j[pb], b->x[pb]); ++pb; }
            else {
                float z = b->x[pb];
                if (op) op(&z, &a->x[pa], &b->x[pb]);
                out_push(&o, a->j[pa], z); ++pa; ++pb;
            }
        }
        o.c.p[i + 1] = o.c.nvals;
    }
    return o.c;
}

static csr_t csr_intersection(int64_t nrows, const csr_t *a, const csr_t *b, GxB_binary_function op) {
    out_t o = out_begin(nrows, a->nvals < b->nvals ? a->nvals : b->nvals);
    for (int64_t i = 0; i < nrows; ++i) {
        int64_t pa = a->p[i], pb = b->p[i];
        const int64_t ea = a->p[i + 1], eb = b->p[i + 1];
        while (pa < ea && pb < eb) {
            if (a->j[pa] < b->j[pb]) ++pa;
            else if (b->j[pb] < a->j[pa]) ++pb;
            else { float z; op(&z, &a->x[pa], &b->x[pb]); out_push(&o, a->j[pa], z); ++pa; ++pb; }
        }
        o.c.p[i + 1] = o.c.nvals;
    }
    return o.c;
}

static csr_t csr_transpose(int64_t nrows, int64_t ncols, const csr_t *a) {
    csr_t t;
    t.p = (int64_t *) xmalloc((size_t) (ncols + 1) * sizeof(int64_t));
    memset(t.p, 0, (size_t) (ncols + 1) * sizeof(int64_t));
    t.j = (int64_t *) xmalloc((size_t) a->nvals * sizeof(int64_t));
    t.x = (float *) xmalloc((size_t) a->nvals * sizeof(float));
    t.nvals = a->nvals;
    for (int64_t q = 0; q < a->nvals; ++q) t.p[a->j[q] + 1]++;
    for (int64_t c = 0; c < ncols; ++c) t.p[c + 1] += t.p[c];
    int64_t *fill = (int64_t *) xmalloc((size_t) (ncols + 1) * sizeof(int64_t));
    memcpy(fill, t.p, (size_t) (ncols + 1) * sizeof(int64_t));
    for (int64_t i = 0; i < nrows; ++i)                       /* rows ascending => columns of the transpose ascending */
        for (int64_t q = a->p[i]; q < a->p[i + 1]; ++q) { const int64_t d = fill[a->j[q]]++; t.j[d] = i; t.x[d] = a->x[q]; }
    free(fill);
    return t;
}

/* ---- objects ---------------------------------------------------------------------------------------- */
static void finish(GrB_Matrix A) {                     /* merge the pending setElement tuples */
    if (!A->npend) return;
    csr_t pend;
    GrB_Info info = csr_from_tuples(&pend, A->nrows, A->ncols, A->pi, A->pj, A->px, A->npend, NULL);
    if (info != GrB_SUCCESS) { fprintf(stderr, "grb_shim: pending tuples out of range\n"); abort(); }
    csr_t merged = csr_union(A->nrows, &A->c, &pend, NULL);
    csr_free(&A->c); csr_free(&pend);
    A->c = merged;
    A->npend = 0;
}

static void write_back(GrB_Matrix C, const GrB_BinaryOp accum, csr_t T) {
    if (!accum) { csr_free(&C->c); C->c = T; return; }
    csr_t Z = csr_union(C->nrows, &C->c, &T, accum->fn);
    csr_free(&C->c); csr_free(&T);
    C->c = Z;
}

GrB_Info GrB_init(GrB_Mode mode) { (void) mode; return GrB_SUCCESS; }
GrB_Info GxB_init(GrB_Mode mode, void *(*m)(size_t), void *(*c)(size_t, size_t), void *(*r)(void *, size_t), void (*f)(void *),
                  bool thread_safe) {
    (void) mode; (void) m; (void) c; (void) r; (void) f; (void) thread_safe;
    return GrB_SUCCESS;
}
GrB_Info GxB_set(GxB_Option_Field field, ...) { (void) field; return GrB_SUCCESS; }     /* the shim is single-threaded */

GrB_Info GrB_finalize(void) {
    const char *prefix = getenv("GRBSHIM_DUMP");
    if (prefix && *prefix) {
        const char *rank = getenv("MPISHIM_RANK");
        char path[4096];
        snprintf(path, sizeof path, "%s.%s", prefix, rank ? rank : "0");
        FILE *f = fopen(path, "wb");
        if (!f) { fprintf(stderr, "grb_shim: cannot write %s\n", path); return GrB_PANIC; }
        int64_t count = 0;
        for (GrB_Matrix A = live_head; A; A = A->next) ++count;
        fwrite("GRBD", 1, 4, f);
        fwrite(&count, sizeof count, 1, f);
        for (GrB_Matrix A = live_head; A; A = A->next) {
            finish(A);
            int64_t hdr[4] = {A->serial, A->nrows, A->ncols, A->c.nvals};
            fwrite(hdr, sizeof(int64_t), 4, f);
            for (int64_t i = 0; i < A->nrows; ++i)
                for (int64_t q = A->c.p[i]; q < A->c.p[i + 1]; ++q) fwrite(&i, sizeof i, 1, f);
            fwrite(A->c.j, sizeof(int64_t), (size_t) A->c.nvals, f);
            fwrite(A->c.x, sizeof(float), (size_t) A->c.nvals, f);
        }
        if (fclose(f) != 0) return GrB_PANIC;
    }
    return GrB_SUCCESS;
}

GrB_Info GrB_UnaryOp_new(GrB_UnaryOp *op, GxB_unary_function fn, GrB_Type z, GrB_Type x) {
    if (!op || !fn) return GrB_NULL_POINTER;
    if (z != GrB_FP32 || x != GrB_FP32) return GrB_DOMAIN_MISMATCH;
    *op = (GrB_UnaryOp) xmalloc(sizeof **op); (*op)->fn = fn;
    return GrB_SUCCESS;
}
GrB_Info GrB_BinaryOp_new(GrB_BinaryOp *op, GxB_binary_function fn, GrB_Type z, GrB_Type x, GrB_Type y) {
    if (!op || !fn) return GrB_NULL_POINTER;
    if (z != GrB_FP32 || x != GrB_FP32 || y != GrB_FP32) return GrB_DOMAIN_MISMATCH;
    *op = (GrB_BinaryOp) xmalloc(sizeof **op); (*op)->fn = fn;
    return GrB_SUCCESS;
}
GrB_Info GrB_Monoid_new_FP32(GrB_Monoid *monoid, GrB_BinaryOp op, float identity) {
    if (!monoid || !op) return GrB_NULL_POINTER;
    *monoid = (GrB_Monoid) xmalloc(sizeof **monoid); (*monoid)->op = op; (*monoid)->identity = identity;
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_new(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols) {
    if (!A) return GrB_NULL_POINTER;
    if (type != GrB_FP32) return GrB_DOMAIN_MISMATCH;
    GrB_Matrix M = (GrB_Matrix) xmalloc(sizeof *M);
    memset(M, 0, sizeof *M);
    M->nrows = (int64_t) nrows; M->ncols = (int64_t) ncols;
    M->c = csr_empty(M->nrows);
    M->serial = next_serial++;
    M->prev = live_tail; M->next = NULL;
    if (live_tail) live_tail->next = M; else live_head = M;
    live_tail = M;
    *A = M;
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_free(GrB_Matrix *A) {
    if (!A || !*A) return GrB_SUCCESS;
    GrB_Matrix M = *A;
    if (M->prev) M->prev->next = M->next; else live_head = M->next;
    if (M->next) M->next->prev = M->prev; else live_tail = M->prev;
    csr_free(&M->c); free(M->pi); free(M->pj); free(M->px); free(M);
    *A = NULL;
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_clear(GrB_Matrix A) {
    if (!A) return GrB_UNINITIALIZED_OBJECT;
    csr_free(&A->c); A->c = csr_empty(A->nrows); A->npend = 0;
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_wait(GrB_Matrix *A) {
    if (!A || !*A) return GrB_NULL_POINTER;
    finish(*A);
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_nvals(GrB_Index *nvals, const GrB_Matrix A) {
    if (!nvals) return GrB_NULL_POINTER;
    if (!A) return GrB_UNINITIALIZED_OBJECT;
    finish(A);
    *nvals = (GrB_Index) A->c.nvals;
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_build_FP32(GrB_Matrix C, const GrB_Index *I, const GrB_Index *J, const float *X, GrB_Index nvals,
                               const GrB_BinaryOp dup) {
    if (!C) return GrB_UNINITIALIZED_OBJECT;
    if (nvals && (!I || !J || !X)) return GrB_NULL_POINTER;
    if (!dup) return GrB_NULL_POINTER;                   /* v1.3: dup is required */
    if (C->c.nvals || C->npend) return GrB_OUTPUT_NOT_EMPTY;
    csr_t T;
    GrB_Info info = csr_from_tuples(&T, C->nrows, C->ncols, (const int64_t *) I, (const int64_t *) J, X, (int64_t) nvals, dup->fn);
    if (info != GrB_SUCCESS) return info;
    csr_free(&C->c); C->c = T;
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_setElement_FP64(GrB_Matrix C, double x, GrB_Index i, GrB_Index j) {
    if (!C) return GrB_UNINITIALIZED_OBJECT;
    if ((int64_t) i >= C->nrows || (int64_t) j >= C->ncols) return GrB_INVALID_INDEX;
    if (C->npend == C->cappend) {
        C->cappend = C->cappend ? 2 * C->cappend : 64;
        C->pi = (int64_t *) xrealloc(C->pi, (size_t) C->cappend * sizeof(int64_t));
        C->pj = (int64_t *) xrealloc(C->pj, (size_t) C->cappend * sizeof(int64_t));
        C->px = (float *) xrealloc(C->px, (size_t) C->cappend * sizeof(float));
    }
    C->pi[C->npend] = (int64_t) i; C->pj[C->npend] = (int64_t) j; C->px[C->npend] = (float) x; C->npend++;
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_extractTuples_FP32(GrB_Index *I, GrB_Index *J, float *X, GrB_Index *nvals, const GrB_Matrix A) {
    if (!nvals) return GrB_NULL_POINTER;
    if (!A) return GrB_UNINITIALIZED_OBJECT;
    finish(A);
    if ((int64_t) *nvals < A->c.nvals) return GrB_INSUFFICIENT_SPACE;
    for (int64_t i = 0; i < A->nrows; ++i)
        for (int64_t q = A->c.p[i]; q < A->c.p[i + 1]; ++q) {
            if (I) I[q] = (GrB_Index) i;
            if (J) J[q] = (GrB_Index) A->c.j[q];
            if (X) X[q] = A->c.x[q];
        }
    *nvals = (GrB_Index) A->c.nvals;
    return GrB_SUCCESS;
}

/* ---- operations ------------------------------------------------------------------------------------- */
static int cmp_i64(const void *a, const void *b) {
    const int64_t u = *(const int64_t *) a, v = *(const int64_t *) b;
    return u < v ? -1 : (u > v);
}

GrB_Info GrB_mxm(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring, const GrB_Matrix A,
                 const GrB_Matrix B, const GrB_Descriptor desc) {
    if (!C || !A || !B || !semiring) return GrB_UNINITIALIZED_OBJECT;
    if (Mask) return GrB_INVALID_VALUE;                  /* main.c never passes a mask; refuse loudly */
    finish(A); finish(B); finish(C);
    const int ta = desc && desc->t0, tb = desc && desc->t1;
    const int64_t am = ta ? A->ncols : A->nrows, ak = ta ? A->nrows : A->ncols;
    const int64_t bk = tb ? B->ncols : B->nrows, bn = tb ? B->nrows : B->ncols;
    if (ak != bk || C->nrows != am || C->ncols != bn) return GrB_DIMENSION_MISMATCH;
    csr_t At, Bt;
    const csr_t *a = &A->c, *b = &B->c;
    if (ta) { At = csr_transpose(A->nrows, A->ncols, &A->c); a = &At; }
    if (tb) { Bt = csr_transpose(B->nrows, B->ncols, &B->c); b = &Bt; }
    float *acc = (float *) xmalloc((size_t) bn * sizeof(float));
    int64_t *mark = (int64_t *) xmalloc((size_t) bn * sizeof(int64_t));
    int64_t *list = (int64_t *) xmalloc((size_t) bn * sizeof(int64_t));
    for (int64_t j = 0; j < bn; ++j) mark[j] = -1;
    out_t o = out_begin(am, a->nvals);
    const int second = semiring->second;
    for (int64_t i = 0; i < am; ++i) {
        int64_t cnt = 0;
        for (int64_t pa = a->p[i]; pa < a->p[i + 1]; ++pa) {         /* k ascending */
            const int64_t k = a->j[pa];
            const float av = a->x[pa];
            for (int64_t pb = b->p[k]; pb < b->p[k + 1]; ++pb) {
                const int64_t j = b->j[pb];
                const float t = second ? b->x[pb] : av * b->x[pb];
                if (mark[j] != i) { mark[j] = i; acc[j] = t; list[cnt++] = j; }
                else acc[j] = acc[j] + t;
            }
        }
        if (cnt == bn) { for (int64_t j = 0; j < bn; ++j) out_push(&o, j, acc[j]); }
        else {
            qsort(list, (size_t) cnt, sizeof(int64_t), cmp_i64);
            for (int64_t q = 0; q < cnt; ++q) out_push(&o, list[q], acc[list[q]]);
        }
        o.c.p[i + 1] = o.c.nvals;
    }
    free(acc); free(mark); free(list);
    if (ta) csr_free(&At);
    if (tb) csr_free(&Bt);
    write_back(C, accum, o.c);
    return GrB_SUCCESS;
}

static GrB_Info ewise(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A,
                      const GrB_Matrix B, const GrB_Descriptor desc, int add) {
    if (!C || !A || !B || !op) return GrB_UNINITIALIZED_OBJECT;
    if (Mask || (desc && (desc->t0 || desc->t1))) return GrB_INVALID_VALUE;
    finish(A); finish(B); finish(C);
    if (A->nrows != B->nrows || A->ncols != B->ncols || C->nrows != A->nrows || C->ncols != A->ncols) return GrB_DIMENSION_MISMATCH;
    csr_t T = add ? csr_union(A->nrows, &A->c, &B->c, op->fn) : csr_intersection(A->nrows, &A->c, &B->c, op->fn);
    write_back(C, accum, T);
    return GrB_SUCCESS;
}
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                      const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
    return ewise(C, Mask, accum, op, A, B, desc, 1);
}
GrB_Info GrB_Matrix_eWiseMult_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                       const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
    return ewise(C, Mask, accum, op, A, B, desc, 0);
}

static csr_t csr_copy_pattern(int64_t nrows, const csr_t *a) {
    csr_t t;
    t.p = (int64_t *) xmalloc((size_t) (nrows + 1) * sizeof(int64_t));
    memcpy(t.p, a->p, (size_t) (nrows + 1) * sizeof(int64_t));
    t.j = (int64_t *) xmalloc((size_t) a->nvals * sizeof(int64_t));
    memcpy(t.j, a->j, (size_t) a->nvals * sizeof(int64_t));
    t.x = (float *) xmalloc((size_t) a->nvals * sizeof(float));
    t.nvals = a->nvals;
    return t;
}

GrB_Info GrB_Matrix_apply(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_UnaryOp op, const GrB_Matrix A,
                          const GrB_Descriptor desc) {
    if (!C || !A || !op) return GrB_UNINITIALIZED_OBJECT;
    if (Mask || (desc && desc->t0)) return GrB_INVALID_VALUE;
    finish(A); finish(C);
    if (C->nrows != A->nrows || C->ncols != A->ncols) return GrB_DIMENSION_MISMATCH;
    csr_t T = csr_copy_pattern(A->nrows, &A->c);
    for (int64_t q = 0; q < T.nvals; ++q) op->fn(&T.x[q], &A->c.x[q]);
    write_back(C, accum, T);
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_apply_BinaryOp2nd_FP32(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op,
                                           const GrB_Matrix A, float y, const GrB_Descriptor desc) {
    if (!C || !A || !op) return GrB_UNINITIALIZED_OBJECT;
    if (Mask || (desc && desc->t0)) return GrB_INVALID_VALUE;
    finish(A); finish(C);
    if (C->nrows != A->nrows || C->ncols != A->ncols) return GrB_DIMENSION_MISMATCH;
    csr_t T = csr_copy_pattern(A->nrows, &A->c);
    for (int64_t q = 0; q < T.nvals; ++q) op->fn(&T.x[q], &A->c.x[q], &y);
    write_back(C, accum, T);
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_assign_FP32(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, float x, const GrB_Index *I,
                                GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc) {
    (void) ni; (void) nj; (void) desc;
    if (!C) return GrB_UNINITIALIZED_OBJECT;
    if (Mask || I != GrB_ALL || J != GrB_ALL) return GrB_INVALID_VALUE;      /* only C(:,:) = x, the form of main.c:414 */
    finish(C);
    out_t o = out_begin(C->nrows, C->nrows * C->ncols);
    for (int64_t i = 0; i < C->nrows; ++i) {
        for (int64_t j = 0; j < C->ncols; ++j) out_push(&o, j, x);
        o.c.p[i + 1] = o.c.nvals;
    }
    write_back(C, accum, o.c);
    return GrB_SUCCESS;
}

GrB_Info GrB_Matrix_reduce_FP32(float *c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A,
                                const GrB_Descriptor desc) {
    (void) desc;
    if (!c) return GrB_NULL_POINTER;
    if (!monoid || !A) return GrB_UNINITIALIZED_OBJECT;
    finish(A);
    float s = monoid->identity;
    if (A->c.nvals) {
        s = A->c.x[0];
        for (int64_t q = 1; q < A->c.nvals; ++q) { float t; monoid->op->fn(&t, &s, &A->c.x[q]); s = t; }
    }
    if (accum) { float t; accum->fn(&t, c, &s); *c = t; } else *c = s;
    return GrB_SUCCESS;
}
