/*
 * pgcn_hip.h -- C ABI of the MI355X (gfx950) aggregation library, libpgcn_hip.so
 *
 * This is the drop-in boundary for the hot path of the reference's GPU engine
 * (/root/reference/GPU/PGCN.py).  The reference has no FFI of its own: its hot
 * path is a handful of PyTorch calls inside PSpMM / communicate_fgm.  Each entry
 * point below replaces one of those call sites (cited per function) and is what
 * a ctypes / cffi / pybind stub in the reference would bind.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*.
 *   - every launch is asynchronous on the given stream; nothing synchronises,
 *     nothing allocates device memory (work-space is passed in), so all calls
 *     are graph-capturable and re-entrant across streams.
 *   - return 0 on success, a negative PGCN_E* code otherwise; the HIP / RCCL
 *     error text is kept per thread in pgcn_last_error().  Never throws.
 *   - fp32 values, int32 column ids, int64 row pointers, row-major dense
 *     panels with a leading dimension in ELEMENTS.
 */
#ifndef PGCN_HIP_H
#define PGCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGCN_ABI_VERSION 3   /* r06: panel_list of pgcn_spmm_dense_bf16x3_f32 holds first ROWS (any origin); pgcn_spmm_dense_f32 and the slice-pair
                              * plan option are gone.  r04: + pgcn_spmm_dense_bf16x3_f32; r03: entry-major de / src of the GAT entry points */

#define PGCN_OK 0
#define PGCN_EINVAL (-1)  /* bad argument (null pointer, misaligned, negative size) */
#define PGCN_EHIP (-2)    /* a HIP runtime call / kernel launch failed              */
#define PGCN_ERCCL (-3)   /* an RCCL call failed                                    */
#define PGCN_ENOMEM (-4)  /* caller-provided capacity / work-space too small        */
#define PGCN_EUNSUPPORTED (-5) /* valid input of a kind this library does not handle  */

typedef void *pgcn_stream_t; /* hipStream_t */

/* flags for the SpMM entry points */
#define PGCN_SPMM_ACCUMULATE 1u  /* C += A.B instead of C = A.B                      */
#define PGCN_SPMM_XCD_SWIZZLE 2u /* unsliced plans only: give each XCD a contiguous row range */
#define PGCN_SPMM_OFFSETS32 4u   /* caller guarantees (max col + 1) * ldb * 4 < 2^32 bytes */
#define PGCN_SPMM_NO_FIXUP 8u    /* plan call: leave partial sums in the work-space; the caller
                                    combines them with pgcn_spmm_fixup_f32                  */
#define PGCN_SPMM_FPASS64 16u   /* gather kernels: 64 features per pass (grid y = passes, pass-major order) */

#define PGCN_MAX_SLICES 8        /* = XCDs of an MI355X */
#define PGCN_MAX_COL_GROUPS 64   /* column groups per slice (time slicing of the column space) */
#define PGCN_CORE_TR 128         /* rows per tile of the LDS-tiled core kernel    */
#define PGCN_CORE_TC 128         /* columns per panel of the LDS-tiled core kernel */
#define PGCN_STRIP_TR 512        /* rows per tile of the strip kernel (pgcn_spmm_strip_f32) */
#define PGCN_STRIP_B 2           /* pair slots per row and strip record                  */

int pgcn_abi_version(void);
const char *pgcn_last_error(void);

/* ---- device query (plumbing for the bench / tests) ---------------------- */
/* out[0]=#CUs, out[1]=wavefront size, out[2]=gcnArch number (950), out[3]=L2 bytes */
int pgcn_device_info(int32_t device, int64_t out[4]);

/* ---- MatrixMarket ingest (host only, multi-threaded) -----------------------------
 * replaces scipy.io.mmread(path_A)             GPU/PGCN.py:171   ("next" row N1)
 * out[0..2] = rows, cols, stored entries; out[3] = bit0 pattern | bit1 symmetric |
 * bit2 skew-symmetric | bit3 integer.  pgcn_mtx_read_coo fills 0-based COO arrays of
 * capacity `cap` (>= 2 x stored entries is always enough); symmetric / skew-symmetric
 * files are expanded like mmread does.  nthreads <= 0: all hardware threads.         */
int pgcn_mtx_info(const char *path, int64_t out[4]);
int pgcn_mtx_read_coo(const char *path, int64_t cap, int64_t *row, int64_t *col, float *val,
                      int64_t *nnz_out, int32_t nthreads);

/* ---- 1D partition, host side (multi-threaded) ----------------------------------------
 * pgcn_build_comm_maps replaces compute_communication_maps(A, partvec, rank, size)
 * GPU/PGCN.py:37-51.  (row, col) are 0-based coordinates of stored entries -- all of them, as the
 * reference scans, or any subset that holds every entry with one end on `rank`.  For every peer
 * q != rank:  send ids [send_off[q], send_off[q+1]) = sorted unique columns j owned by `rank` that
 * appear in a row owned by q (PGCN.py:44-47);  recv ids [recv_off[q], recv_off[q+1]) = sorted
 * unique columns owned by q that appear in a row owned by `rank` (:41-43).  The own-rank segments
 * are empty (:49-50).  Offsets arrays have nranks+1 entries.  Sizing call: pass send_ids =
 * recv_ids = NULL, read the totals from send_off[nranks] / recv_off[nranks].
 *
 * pgcn_load_mtx_partition replaces mmread (:171) + get_partitiont_of_adjacency_matrix (:53-64):
 * the stored entries of the rows owned by `rank`, GLOBAL 0-based coordinates, file order.
 * Sizing call: row = col = val = NULL.                                                        */
int pgcn_build_comm_maps(const int64_t *row, const int64_t *col, int64_t nnz, const int32_t *partvec,
                         int64_t n, int32_t rank, int32_t nranks, int64_t *send_off,
                         int64_t *recv_off, int64_t *send_ids, int64_t cap_send, int64_t *recv_ids,
                         int64_t cap_recv, int32_t nthreads);
int pgcn_load_mtx_partition(const char *path, const int32_t *partvec, int64_t n, int32_t rank,
                            int64_t cap, int64_t *row, int64_t *col, float *val, int64_t *nnz_out,
                            int32_t nthreads);

/* ---- binary CSR shards (host only; "next" row N1, papers100M-scale ingest) ---------------------
 * One file per rank with ONLY that rank's rows, in the layout the engine consumes: replaces the text parse
 * of the whole matrix on every rank (GPU/PGCN.py:171 mmread, :37-64) and lifts the 32-bit pin arrays of the
 * reference's partitioner front-end (GCN-HP/main.cpp:286-307).  Little endian:
 *   8 x int64 {magic "PGCSR001", n_global, nrows, nnz, rank, nparts, flags = 0, 0}
 *   int64 rows[nrows] (global ids, ascending) | int64 rowptr[nrows + 1] | int32 col[nnz] (global ids,
 *   padded to 8 bytes) | fp32 val[nnz].
 * pgcn_shard_info: out = {n_global, nrows, nnz, rank, nparts, flags}.  pgcn_shard_read fills caller arrays
 * (capacities in elements) and validates the structure.                                                  */
int pgcn_shard_write(const char *path, int64_t n_global, int32_t rank, int32_t nparts, int64_t nrows,
                     const int64_t *rows, const int64_t *rowptr, const int32_t *col, const float *val);
int pgcn_shard_info(const char *path, int64_t out[6]);
int pgcn_shard_read(const char *path, int64_t cap_rows, int64_t cap_nnz, int64_t *rows, int64_t *rowptr,
                    int32_t *col, float *val);

/* ---- CSR SpMM ------------------------------------------------------------
 * C[nrows x f] (+)= A[nrows x *] . B[* x f]
 * replaces  torch.sparse.mm(A, H)            GPU/PGCN.py:127
 *      and  torch.sparse.mm(A.t(), grad)     GPU/PGCN.py:132 (pass the CSR of A^T)
 *      ==   GrB_mxm(AH, PLUS_TIMES_FP32, A, H)  Parallel-GCN/main.c:271,295,376,400
 * val may be NULL (pattern matrix, all ones).  One task per row, no work-space.
 */
int pgcn_spmm_csr_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                      int64_t nrows, const float *B, int64_t ldb, float *C, int64_t ldc,
                      int32_t f, uint32_t flags, pgcn_stream_t stream);

/* Load-balanced, XCD-sliced variant driven by a plan (see pgcn_spmm_plan_host).
 * The entries of each row must be stored grouped by slice = col % nslices when the
 * plan was built with nslices > 1; task segment s (seg[s]..seg[s+1], HOST array of
 * nslices+1 entries) is executed by the workgroups with blockIdx % nslices == s, which
 * the dispatcher places on XCD s: every private 4 MiB L2 then caches a disjoint 1/8 of
 * the rows of B (placement only affects speed, never results).  Rows with several
 * tasks combine their partial sums through `partial_ws` (capacity partial_ws_elems >=
 * nslots * f floats) in a fixed order in a second kernel -- deterministic, no atomics
 * on C.  row_map (optional, device) maps CSR row r to output row row_map[r]: the
 * row-subset form used for the halo pass of the local/halo split.                  */
int pgcn_spmm_csr_plan_f32(const int64_t *rowptr, const int32_t *col, const float *val,
                           const int32_t *tasks, int64_t ntasks, const int64_t *seg,
                           int32_t nslices, const int32_t *fix, int64_t nfix,
                           const int32_t *row_map, const float *B, int64_t ldb, float *C,
                           int64_t ldc, int32_t f, float *partial_ws, int64_t partial_ws_elems,
                           int64_t nslots, uint32_t flags, pgcn_stream_t stream);

/* Host-side plan builder (pure CPU, no HIP).  rowptr_host: nrows+1 entries;
 * slice_cnt: nrows x (nslices*ngroups) entry counts per (row, slice, column group), index
 *        s*ngroups + g, or NULL when nslices*ngroups == 1.  ngroups > 1 additionally orders
 *        every segment column-group-major (see the plan description in csrc/pgcn_core.cpp);
 *        rows with fewer than group_min_row entries keep one piece per slice.
 * tasks: 4 x int32 per task {kbeg low, kbeg high, length, dst}: absolute offset of the
 *        first entry, number of entries, dst >= 0 partial slot / dst < 0 direct row ~dst;
 *        grouped by slice, longest first; seg (out, nslices+1) = segment boundaries.
 * fix:   4 x int32 per multi-task row {row, first slot, #tasks, 0}
 * Rows with <= small_row entries are not sliced (one direct task each).
 * row_flags (optional, nrows bytes): a non-zero flag marks a row that also receives
 * partial sums from elsewhere (the core kernel): it never writes C directly -- even a
 * single task gets a slot and a fix record -- and gets no task at all when it is empty.
 * Call with tasks == NULL to obtain the counts, then again with buffers.      */
int pgcn_spmm_plan_host(const int64_t *rowptr_host, const int32_t *slice_cnt,
                        const uint8_t *row_flags, int64_t nrows,
                        int32_t nslices, int32_t ngroups, int32_t group_min_row, int32_t chunk,
                        int32_t small_row, int32_t *tasks,
                        int64_t cap_tasks, int32_t *fix, int64_t cap_fix, int64_t *seg,
                        int64_t *ntasks, int64_t *nfix, int64_t *nslots);
/* ... with placement options.  plan_flags = 0: exactly pgcn_spmm_plan_host. */
#define PGCN_PLAN_AFFINE_SMALL 0x40000000   /* plan_flags: an unsliced row's task runs on the segment (XCD) of its fullest slice */
int pgcn_spmm_plan_host_ex(const int64_t *rowptr_host, const int32_t *slice_cnt,
                           const uint8_t *row_flags, int64_t nrows,
                           int32_t nslices, int32_t ngroups, int32_t group_min_row, int32_t chunk,
                           int32_t small_row, int32_t plan_flags, int32_t *tasks,
                           int64_t cap_tasks, int32_t *fix, int64_t cap_fix, int64_t *seg,
                           int64_t *ntasks, int64_t *nfix, int64_t *nslots);

/* ---- LDS-tiled dense core ------------------------------------------------------
 * Same product, for the entries that fall into DENSE 128 x 128 tiles of the block (after
 * the rows/columns were relabelled by decreasing degree): a workgroup stages the 128
 * feature rows of a column panel in LDS once and serves all entries of its 128 matrix
 * rows from LDS (~4x the L1 gather rate, L2 traffic / fill factor).  Layout, all device:
 *   work       4 x int32 per piece {tile row index, first tile, one-past-last tile, first slot}
 *   tile_panel int32 per dense tile: panel index (column block)
 *   tile_base  int64 per dense tile: offset of its entries in ccol / cval
 *   seg_off    int32 (TR+1) per dense tile: segment bounds of the tile's rows, rows in
 *              group-major order (index g*8+j  <->  row j*16+g of the tile)
 *   ccol,cval  column inside the panel (0..127) and value of every core entry
 * A piece writes TR partial rows to slots [first slot, first slot + TR) of partial_ws.   */
int pgcn_spmm_core_f32(const int32_t *work, int64_t nwork, const int32_t *tile_panel,
                       const int64_t *tile_base, const int32_t *seg_off, const int32_t *ccol,
                       const float *cval, const float *B, int64_t ldb, int64_t ncols, int32_t f,
                       float *partial_ws, int64_t partial_ws_elems, int64_t nslots_total,
                       pgcn_stream_t stream);

/* ---- multi-head weighted SpMM (GAT, "next" row N3) --------------------------------------------
 *   C[i, k*d .. (k+1)*d) (+)= sum over the stored entries e of row i:  alpha[k * plane_stride + e] * B[col[e], k*d ..]
 * for all `heads` heads in ONE launch: replaces `attention @ Z` of GPU/PGAT.py:148 on the stored entries
 * (the reference multiplies a dense n x n attention matrix).  alpha: head-major planes in the storage
 * order of `col`.  tasks / seg / nslices / fix / nslots: the plan of pgcn_spmm_plan_host on the same
 * structure (tasks == NULL: one task per row, nrows rows).  One wave gathers a whole heads * d wide row.
 * Supported: heads <= 8, heads * d <= 256, d % 4 == 0, ldb % 4 == 0, ldc % 4 == 0, 16-byte aligned
 * B / C / work-space; otherwise PGCN_EUNSUPPORTED (run one pgcn_spmm_csr_plan_f32 per head instead). */
int pgcn_spmm_heads_f32(const int64_t *rowptr, const int32_t *col, const float *alpha, int64_t plane_stride,
                        int32_t heads, int32_t d, int64_t nrows, const int32_t *tasks, int64_t ntasks,
                        const int64_t *seg, int32_t nslices, const int32_t *fix, int64_t nfix,
                        const float *B, int64_t ldb, float *C, int64_t ldc, float *partial_ws,
                        int64_t partial_ws_elems, int64_t nslots, uint32_t flags, pgcn_stream_t stream);

/* ---- strip tiles: 512 x 128, LDS-staged, asynchronous double-buffered pipeline -------------
 * Same product (torch.sparse.mm, GPU/PGCN.py:127,132) for the entries of TALL tiles: a 1024-thread
 * workgroup stages the 128 feature rows of a column panel in LDS once (global_load_lds, one run of
 * records ahead) and serves all entries of its 512 matrix rows from LDS; pays from a few hundred
 * entries per tile on.
 *   work  4 x int32 per piece {tile row, first record, one-past-last record, first slot}
 *   recs  4 x int32 per record {panel, flags, next panel, layer}; flags bit 0: the panel is the one
 *         of the previous record of the piece; `next panel`: on a record that starts a run of one
 *         panel, the panel of the piece's NEXT run (-1: none), else -1.  A record is one LAYER of a
 *         tile: the (2 l)-th and (2 l + 1)-th stored entry (column order) of every one of its 512 rows
 *   pairs 512 x 2 int32 pairs per record {byte offset of the column's row in the staged panel
 *         (= column-in-panel * 512), value bits}, rows in (group, row slot) order with local row =
 *         row slot * 64 + group (64 groups of 16 lanes, 8 row slots); an unused slot holds
 *         {65536 (an all-zero LDS row), 0.0f}
 * The staged panel of column block p is the 128 rows of B starting at p * 128 -- except the last
 * block of an operand with ncols % 128 != 0, which is the window [ncols - 128, ncols) (offsets are
 * relative to the window; ncols >= 128 required).
 * A piece leaves 512 partial rows in slots [first slot, first slot + 512) of partial_ws for
 * pgcn_spmm_fixup_f32.  All three arrays 16-byte aligned.  f % 4 == 0 with 16-byte aligned B /
 * partial_ws takes the LDS pipeline (128 features per workgroup); anything else a plain kernel. */
int pgcn_spmm_strip_f32(const int32_t *work, int64_t nwork, const int32_t *recs, const int32_t *pairs,
                        const float *B, int64_t ldb, int64_t ncols, int32_t f, float *partial_ws,
                        int64_t partial_ws_elems, int64_t nslots_total, pgcn_stream_t stream);


/* The densest 512 x 128 BLOCKS through the bf16 matrix cores at fp32 accuracy (v_mfma_f32_32x32x16_bf16; every fp32
 * operand is the exact sum of three bf16 numbers, the six partial products that matter are accumulated in fp32,
 * smallest first: the error class of an fp32 dot product, deterministic).  Part of PSpMM's torch.sparse.mm
 * (/root/reference/GPU/PGCN.py:127,132) like the other SpMM entry points.
 *   vals3[block][w][unit = 2 ks + rb][h][lane][e] = A[64 w + 32 rb + (lane & 31)][16 ks + 8 (lane >> 5) + 4 h + e]
 *     (65 536 fp32 per block, 16-byte aligned: the A-operand order of the instruction for wave w of 8);
 *   work: 4 x int32 per piece {block row, first block, number of blocks, first slot}; a piece leaves a 512 x f block of
 *     partial sums in partial_ws (slot rows of f floats) for pgcn_spmm_fixup_f32;
 *   panel_list[npanels]: FIRST ROW p of the distinct panels the blocks refer to (rows [p, p + 128) of B -- any p since ABI 3, not
 *     only multiples of 128 -- rows >= ncols read as zero), blk_img[b]: position of block b's panel in panel_list.
 * The call first splits the listed panels of B into bf16 planes in image_ws (pgcn_dense_bf16x3_image_bytes(npanels, f)
 * bytes, 16-byte aligned; scratch, rewritten by every call), then runs the blocks.  Zeros of a block are structural:
 * a panel holding Inf / NaN takes an exact (slow) path that multiplies only where A != 0.  Any f, ldb, alignment of B. */
int64_t pgcn_dense_bf16x3_image_bytes(int64_t npanels, int32_t f);
int pgcn_spmm_dense_bf16x3_f32(const int32_t *work, int64_t nwork, const int32_t *blk_img, const float *vals3,
                               const int32_t *panel_list, int64_t npanels, const float *B, int64_t ldb, int64_t ncols,
                               int32_t f, void *image_ws, int64_t image_ws_bytes, float *partial_ws,
                               int64_t partial_ws_elems, int64_t nslots_total, pgcn_stream_t stream);

/* C[row] (+)= sum of the partial-sum slots listed for the row, in list order.
 * fix: 4 x int32 per row {row, begin, count, 0}; slot_ids (optional): the row's slots are
 * slot_ids[begin .. begin+count), or begin .. begin+count when slot_ids is NULL.        */
int pgcn_spmm_fixup_f32(const int32_t *fix, int64_t nfix, const int32_t *slot_ids,
                        const int32_t *row_map, const float *partial_ws, float *C, int64_t ldc,
                        int32_t f, uint32_t flags, pgcn_stream_t stream);

/* The same product with the weights RECOMPUTED per entry instead of read from planes -- the transposed product of
 * the GAT backward pass, dZ = A_alpha^T . dOut on the transposed structure (rows = columns of A): entry (row j,
 * col i) weighs alpha_ij = (exp(e - m_i) - em_i) / D_i, e = [LeakyReLU](s1_i + s2_j), from rowstat[i] = (s1, m, 1/D, em)
 * (the softmax's per-row statistics, [ncols x heads x 4], 16-byte aligned) and s2[j] ([nrows x lds2]).  The
 * arithmetic of pgcn_gat_edge_weights_t_f32 inside the gather kernel, bit for bit: the alpha^T planes and the
 * walk that wrote them disappear (GPU/PGAT.py:148 backward).  mode / slope as in the attention kernels below.   */
int pgcn_spmm_heads_recompute_f32(const int64_t *rowptr, const int32_t *col, const float *rowstat,
                                  const float *s2, int64_t lds2, float slope, int32_t mode, int32_t heads,
                                  int32_t d, int64_t nrows, const int32_t *tasks, int64_t ntasks,
                                  const int64_t *seg, int32_t nslices, const int32_t *fix, int64_t nfix,
                                  const float *B, int64_t ldb, float *C, int64_t ldc, float *partial_ws,
                                  int64_t partial_ws_elems, int64_t nslots, uint32_t flags,
                                  pgcn_stream_t stream);

/* ... and with the EDGE GRADIENT of the same pass (r03).  The backward of GPU/PGAT.py:144-149 needs, per stored (i, j),
 * dp_ij = <dOut_i, Z_j> and alpha_ij dOut_i summed into dZ_j: the same pairs of rows.  On the transposed structure Z_j is
 * the task's own row (Z, [nrows x ldz]) and dOut_i (B) is gathered anyway, so one pass yields
 *   C[j, 0 .. F)            (+)= sum_i alpha_ij dOut_i                       (as pgcn_spmm_heads_recompute_f32)
 *   de[q, k]                 = (alpha_ij + beta_i)(dp_ij - t_i)[x LeakyReLU'(s1_i + s2_j)]   entry-major [nnz x heads],
 *                              q = the entry's position in THIS (transposed) structure; t = [ncols x heads];
 *                              de = NULL: not kept (see pgcn_spmm_heads_forward2_f32)
 *   C[j, F .. F + heads)    (+)= sum_i de_ij  (= ds2_j; columns up to the next multiple of 4 are written as zero)
 * so ldc >= F + heads rounded up to 4, and a partial row is that wide.  ds1 = the column sums of de:
 * pgcn_csr_row_sums_f32 over the forward structure with the inverse permutation.  PGCN_EUNSUPPORTED unless
 * d is 32, 64, 128 or 256 (and the limits of pgcn_spmm_heads_f32): callers then run pgcn_gat_edge_grad_*_f32 and
 * pgcn_spmm_heads_recompute_f32.                                                                              */
int pgcn_spmm_heads_grad_f32(const int64_t *rowptr, const int32_t *col, const float *rowstat, const float *s2,
                             int64_t lds2, float slope, int32_t mode, int32_t heads, int32_t d, int64_t nrows,
                             const int32_t *tasks, int64_t ntasks, const int64_t *seg, int32_t nslices,
                             const int32_t *fix, int64_t nfix, const float *B, int64_t ldb, const float *Z,
                             int64_t ldz, const float *t, float *C, int64_t ldc, float *de, float *partial_ws,
                             int64_t partial_ws_elems, int64_t nslots, uint32_t flags, pgcn_stream_t stream);

/* The FORWARD product with recomputed weights and a second accumulator (r03): out = A_alpha . B as pgcn_spmm_heads_f32, but
 * alpha_ij is recomputed from rowstat[i] (the task's own row, [nrows x heads x 4]) and s2[col] ([ncols x lds2]) -- no alpha
 * planes, pgcn_gat_edge_softmax_f32 may be called with alpha = NULL -- and the same gathered rows also give
 *   C2[i, 0 .. F)          (+)= V_i = sum_j c_ij B_j,   c_ij = alpha_ij LeakyReLU'(s1_i + s2_j) (mode 0) | alpha_ij + beta_i (mode 1)
 *   C2[i, F .. F + heads)  (+)= C_i = sum_j c_ij        (columns up to the next multiple of 4 are written as zero)
 * so that the backward pass gets ds1_i = sum_j de_ij = <dOut_i, V_i> - t_i C_i from row-local data and never needs the
 * per-entry gradient (pgcn_spmm_heads_grad_f32 with de = NULL).  ldc2 >= F + heads rounded up to 4; the work-space holds
 * nslots x (2 F + that) floats.  PGCN_EUNSUPPORTED unless d is 32, 64, 128 or 256 (GPU/PGAT.py:144-149).            */
int pgcn_spmm_heads_forward2_f32(const int64_t *rowptr, const int32_t *col, const float *rowstat, const float *s2,
                                 int64_t lds2, float slope, int32_t mode, int32_t heads, int32_t d, int64_t nrows,
                                 const int32_t *tasks, int64_t ntasks, const int64_t *seg, int32_t nslices,
                                 const int32_t *fix, int64_t nfix, const float *B, int64_t ldb, float *C, int64_t ldc,
                                 float *C2, int64_t ldc2, float *partial_ws, int64_t partial_ws_elems, int64_t nslots,
                                 uint32_t flags, pgcn_stream_t stream);

/* The dense 512 x 128 BLOCKS of the attention pattern on the bf16 matrix cores (r06, pgcn_gat_blocks.hip): the entries inside the
 * listed blocks of the products above, with the weights computed in registers from the same statistics and split into three bf16
 * planes (fp32 accuracy, as pgcn_spmm_dense_bf16x3_f32).  work / blk_img / panel_list / npanels / image_ws: the block structure and
 * work-space of pgcn_spmm_dense_bf16x3_f32 (image of heads * d features); work_row0[nwork]: first matrix row of every piece;
 * bits: the PATTERN of the blocks, 16 bytes per (block, wave w of 8, lane of 64) in that order -- byte u = 2 ks + rb, bit e = position
 * (row 64 w + 32 rb + lane % 32, column 16 ks + 8 (lane / 32) + e) of the block.  mode 0 (LeakyReLU + softmax) only, d = 64,
 * heads * d <= 256; PGCN_EUNSUPPORTED otherwise (callers keep every entry in the gather kernels).  The calls write SLOT rows only; the
 * caller adds them to the outputs with pgcn_spmm_fixup_f32 (PGCN_SPMM_ACCUMULATE) after the gather kernel of the remaining entries:
 *   forward  (rowstat of the rows [nrows x heads x 4], s2 of the columns, B = Z):  partial_ws = nslots x F floats (out), then
 *            nslots x (F + heads rounded up to 4) floats (V | C | 0): the two outputs of pgcn_spmm_heads_forward2_f32;
 *   backward (the transposed pattern: rowstat and t of the COLUMNS, s2 and Z of the ROWS, B = dOut):  partial_ws = nslots x
 *            (F + heads rounded up to 4) floats (dZ | ds2 | 0): the output row of pgcn_spmm_heads_grad_f32 (de is not available).
 * A non-finite value in a listed panel of B reaches every row of the blocks over that panel (0 x inf inside the MFMA), also rows that do
 * not reference it -- unlike the gather kernels and unlike pgcn_spmm_dense_bf16x3_f32, which redoes such pieces exactly.
 * GPU/PGAT.py:144-149 and its autograd.                                                                                        */
int pgcn_gat_blocks_forward_f32(const int32_t *work, int64_t nwork, const int32_t *work_row0, const int32_t *blk_img,
                                const uint32_t *bits, const int32_t *panel_list, int64_t npanels, const float *rowstat,
                                const float *s2, int64_t lds2, float slope, int32_t heads, int32_t d, int64_t nrows,
                                int64_t ncols, const float *B, int64_t ldb, void *image_ws, int64_t image_ws_bytes,
                                float *partial_ws, int64_t partial_ws_elems, int64_t nslots, pgcn_stream_t stream);
int pgcn_gat_blocks_backward_f32(const int32_t *work, int64_t nwork, const int32_t *work_row0, const int32_t *blk_img,
                                 const uint32_t *bits, const int32_t *panel_list, int64_t npanels, const float *rowstat,
                                 const float *s2, int64_t lds2, const float *t, const float *Z, int64_t ldz, float slope,
                                 int32_t heads, int32_t d, int64_t nrows, int64_t ncols, const float *B, int64_t ldb,
                                 void *image_ws, int64_t image_ws_bytes, float *partial_ws, int64_t partial_ws_elems,
                                 int64_t nslots, pgcn_stream_t stream);

/* ---- GAT path: attention over the stored entries ----------------------
 * Replaces the dense n x n arithmetic of PGAT.forward, GPU/PGAT.py:138-151.  Per head k with
 * s1 = Z a1, s2 = Z a2 (:141-142):  raw_ij = s1[i,k] + s2[col,k]  (:144).
 *   mode 0 (standard GAT): e = LeakyReLU(raw, slope); alpha_ij = softmax over row i's entries.
 *   mode 1 (reference-literal, :145-147): all n_global columns take part, non-edges with logit 0:
 *     m = max(0, max e), D = sum_edges exp(e-m) + (n_global-deg) exp(-m),
 *     alpha_ij = (exp(e_ij-m) - exp(-m))/D, beta[i,k] = exp(-m)/D, so that
 *     out_i = sum_edges alpha_ij Z_j + beta_i sum_all Z_j  (:149).  beta: [nrows x heads].
 * alpha is head-major [heads][nnz] in the storage order of `col`: plane k is the `val` array of
 * pgcn_spmm_csr(_plan)_f32, which does the aggregation out[:,k] = A_alpha_k . Z[:,k].  de (the edge gradient)
 * is ENTRY-major [nnz][heads] in the same order: its only consumer reads it through a permutation (ds2 below),
 * and the heads of an entry are then ONE gather of 4 * heads bytes instead of `heads` random 4-byte gathers.
 * Row lists: rows_wave (NULL = rows 0..nrows_wave-1) get one 64-lane wave each, rows_block one
 * 256-thread workgroup each (hub rows); a row must be in exactly one list to be processed.
 *
 * pgcn_gat_edge_grad_f32: de_ij = (alpha_ij + beta_i)(<dOut[i,k,:], Z[col,k,:]> - t[i,k]) [x LeakyReLU'(raw)],
 * ds1[i,k] = sum_j de_ij; t[i,k] = <dOut[i,k,:], out[i,k,:]> is supplied by the caller.  d = head width.
 * pgcn_csr_row_sums_f32: out[i*ldo + k] = sum over row i's entries p of src[idx(p) * planes + k], idx(p) =
 * perm[p] (NULL = identity), src ENTRY-major [nnz][planes] (the layout of de): with the transposed structure
 * and its permutation this is ds2_j = sum_i de_ij.
 * pgcn_csr_permute_f32: dst[k][p] = src[k][perm[p]] (values of A^T from values of A).
 * rowstat (optional output of the softmax, [nrows x heads x 4] fp32, 16-byte aligned) = (s1, m, 1/D,
 * exp(-m) or 0) per row and head (with rowstat given, alpha may be NULL: statistics only);
 * pgcn_gat_edge_weights_t_f32 recomputes from it the alpha planes
 * in the storage order of the TRANSPOSED structure (rowptr_t/col_t; s2 indexed by its rows) --
 * the same numbers as permuting alpha, without the 4-byte random gathers.
 * All sums run in a fixed order: bit-reproducible, no atomics.                                  */
int pgcn_gat_edge_softmax_f32(const int64_t *rowptr, const int32_t *col, int64_t nrows, int64_t nnz,
                              const int32_t *rows_wave, int64_t nrows_wave, const int32_t *rows_block,
                              int64_t nrows_block, const float *s1, int64_t lds1, const float *s2,
                              int64_t lds2, int32_t heads, float slope, int32_t mode, int64_t n_global,
                              float *alpha, float *beta, float *rowstat, pgcn_stream_t stream);


int pgcn_gat_edge_weights_t_f32(const int64_t *rowptr_t, const int32_t *col_t, int64_t nrows_t,
                                int64_t nnz, const int32_t *rows_wave, int64_t nrows_wave,
                                const int32_t *rows_block, int64_t nrows_block, const float *s2,
                                int64_t lds2, const float *rowstat, int32_t heads, float slope,
                                int32_t mode, float *alpha_t, pgcn_stream_t stream);
int pgcn_gat_edge_grad_f32(const int64_t *rowptr, const int32_t *col, int64_t nrows, int64_t nnz,
                           const int32_t *rows_wave, int64_t nrows_wave, const int32_t *rows_block,
                           int64_t nrows_block, const float *s1, int64_t lds1, const float *s2,
                           int64_t lds2, const float *alpha, const float *beta, const float *Z,
                           int64_t ldz, const float *dOut, int64_t ldo, const float *t, int32_t heads,
                           int32_t d, float slope, int32_t mode, float *de, float *ds1,
                           pgcn_stream_t stream);
/* The edge gradient on an XCD-sliced structure (a row's entries grouped by col % 8, as the SpMM
 * kernels store it): slice_off[i*9 + s] = offset of row i's slice s inside the row (slice_off[i*9+8]
 * = row length).  Workgroup b works on slice b % 8 = its XCD, so each L2 sees one eighth of Z.
 * ds1_slices: [nrows x 8 x heads] partial sums, ds1 = their sum over the 8 slices.  rows (NULL =
 * 0..nlist-1): the rows to process.  Needs heads*d <= 256, d a power of two >= 4, aligned panels. */
int pgcn_gat_edge_grad_sliced_f32(const int64_t *rowptr, const int32_t *col, const int32_t *slice_off,
                                  int64_t nrows, int64_t nnz, const int32_t *rows, int64_t nlist,
                                  const float *s1, int64_t lds1, const float *s2, int64_t lds2,
                                  const float *alpha, const float *beta, const float *Z, int64_t ldz,
                                  const float *dOut, int64_t ldo, const float *t, int32_t heads,
                                  int32_t d, float slope, int32_t mode, float *de, float *ds1_slices,
                                  pgcn_stream_t stream);
/* The edge gradient over the TASKS of the SpMM plan of the same structure (pgcn_spmm_plan_host: XCD slices,
 * chunks of long rows, longest first) instead of one wave per (row, slice): the hub rows of a power-law graph
 * no longer form a tail of a few very long waves.  ds1 [nrows x heads] is complete on return (stream order):
 * single-task rows are written directly, the others go through `nslots` slots of `heads` floats in partial_ws
 * and pgcn_spmm_fixup_f32 (fixed order).  Same shape limits as the sliced variant.                       */
int pgcn_gat_edge_grad_tasks_f32(const int64_t *rowptr, const int32_t *col, int64_t nrows, int64_t nnz,
                                 const int32_t *tasks, int64_t ntasks, const int64_t *seg, int32_t nslices,
                                 const int32_t *fix, int64_t nfix, const float *s1, int64_t lds1,
                                 const float *s2, int64_t lds2, const float *alpha, const float *beta,
                                 const float *Z, int64_t ldz, const float *dOut, int64_t ldo, const float *t,
                                 int32_t heads, int32_t d, float slope, int32_t mode, float *de, float *ds1,
                                 float *partial_ws, int64_t partial_ws_elems, int64_t nslots,
                                 pgcn_stream_t stream);
int pgcn_csr_row_sums_f32(const int64_t *rowptr, const int64_t *perm, int64_t nrows, int64_t nnz,
                          const int32_t *rows_wave, int64_t nrows_wave, const int32_t *rows_block,
                          int64_t nrows_block, const float *src, int32_t planes, float *out,
                          int64_t ldo, pgcn_stream_t stream);

/* The per-row, per-head dot products of the GAT backward in ONE pass (r05): t[i][k] = <dOut_i, out_i> over head k's d columns
 * (the softmax backward of GPU/PGAT.py:147-148 needs it per row), and, with VC = [V | C] (n x (heads*d + heads), ldv: the second
 * accumulator of pgcn_spmm_heads_forward2_f32), ds1[i][k] = <dOut_i, V_i>_k - t[i][k] * C[i][k]; VC == NULL: t only.  t, ds1:
 * n x heads, contiguous.  PGCN_EUNSUPPORTED unless heads * d <= 256, d / 4 a power of two, 16-byte rows. */
int pgcn_gat_row_dots_f32(const float *dOut, int64_t ldo, const float *out, int64_t ldout, const float *VC, int64_t ldv,
                          int64_t n, int32_t heads, int32_t d, float *t, float *ds1, pgcn_stream_t stream);
int pgcn_csr_permute_f32(const float *src, const int64_t *perm, int64_t nnz, int32_t planes,
                         float *dst, pgcn_stream_t stream);

/* ---- loss of the training loop (plumbing next to the graded path, one pass each way) -----------------
 * loss_rows[i] = logsumexp_j X[i,j] - X[i, labels[i]]   = nll_loss(log_softmax(logits), labels) per row,
 * GPU/PGCN.py:214-215 (the caller sums / scales); lse_rows[i] is kept for the backward:
 * dX[i,j] = gscale_dev[0] * scale * (exp(X[i,j] - lse_rows[i]) - [j == labels[i]])   (gscale_dev NULL = 1).
 * f <= 1024 columns; labels int64 in [0, f).                                                             */
int pgcn_nll_rows_f32(const float *X, int64_t ldx, const int64_t *labels, int64_t nrows, int32_t f,
                      float *loss_rows, float *lse_rows, pgcn_stream_t stream);
int pgcn_nll_rows_backward_f32(const float *X, int64_t ldx, const int64_t *labels, const float *lse_rows,
                               const float *gscale_dev, float scale, int64_t nrows, int32_t f, float *dX,
                               int64_t lddx, pgcn_stream_t stream);

/* ---- boundary-row pack / unpack -------------------------------------------
 * out[r,:] = H[idx[r],:]                      replaces H[indices]   GPU/PGCN.py:104
 * H[idx[r],:] (+)= in[r,:]                    replaces X[indices] = buf   :115
 *                                             (accumulate: main.c:295,400 semantics)
 * accumulate != 0 adds with atomics: repeated indices (a boundary row that comes back from several peers)
 * are all added; only the ORDER of the adds of a repeated index is not fixed.  accumulate == 0 with a
 * repeated index keeps one of the rows (like the reference's assignment, quirk Q3).                 */
int pgcn_gather_rows_f32(const float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                         float *out, int64_t ldo, int32_t f, pgcn_stream_t stream);
/* scatter: accumulate = 0 assigns (duplicate ids: the LAST one wins, like `X[indices] = buf`, PGCN.py:115);
 * accumulate = 1 ADDS with atomic fp32 adds: correct for duplicate ids, but the order of the additions to one
 * row -- hence the last bits of the sum -- is NOT reproducible from run to run.  The engine's reverse exchange
 * therefore does not use it: received partial rows are added by a pattern SpMM in fixed order (engine.py). */
int pgcn_scatter_rows_f32(float *H, int64_t ldh, const int32_t *idx, int64_t nidx,
                          const float *in, int64_t ldi, int32_t f, int32_t accumulate,
                          pgcn_stream_t stream);

/* ---- boundary-row exchange over RCCL (xGMI) --------------------------------
 * One all-to-all-v: ncclGroupStart; ncclSend/ncclRecv per peer; ncclGroupEnd.
 * replaces the 2.(P-1) blocking dist.send / dist.recv of GPU/PGCN.py:99-115
 *      ==  the MPI_Isend / MPI_Irecv / MPI_Waitany loop of main.c:238-299.
 * send_off / recv_off: HOST arrays of nranks+1 row offsets into the slabs
 * (rows of `f` floats); the own-rank segment must be empty.                   */
int pgcn_comm_unique_id(void *id128); /* writes 128 bytes (ncclUniqueId) */
int pgcn_comm_init(void **comm, const void *id128, int32_t nranks, int32_t rank);
int pgcn_comm_destroy(void *comm);
int pgcn_exchange_alltoallv_f32(void *comm, const float *send, const int64_t *send_off,
                                float *recv, const int64_t *recv_off, int32_t f,
                                pgcn_stream_t stream);
/* sum-all-reduce of a flat fp32 buffer (all layers' weight gradients fused in one
 * call): replaces the L x dist.all_reduce of GPU/PGCN.py:150-154, main.c:425.  */
int pgcn_allreduce_sum_f32(void *comm, float *buf, int64_t count, pgcn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PGCN_HIP_H */
