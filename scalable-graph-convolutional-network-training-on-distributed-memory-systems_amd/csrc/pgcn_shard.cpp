// pgcn_shard.cpp -- binary CSR shards: one file per rank holding ONLY that rank's rows.
//
// The reference parses the whole MatrixMarket text on every rank (GPU/PGCN.py:171 `mmread`, a Python loop
// over every stored entry at :37-64) and its partitioner front-end indexes pins with 32-bit ints
// (GCN-HP/main.cpp:286-307 `int *xpins`): neither reaches ogbn-papers100M (111 M vertices, 1.6 G entries).
// A shard is the rank's row block in the layout the engine consumes -- int64 row pointers end to end,
// int32 GLOBAL column ids (n < 2^31), fp32 values -- read with one sequential pass, no text, no global COO:
//
//   header   8 x int64 { magic "PGCSR001", n_global, nrows, nnz, rank, nparts, flags (0), reserved (0) }
//   rows     int64 [nrows]      global ids of the owned rows, ascending
//   rowptr   int64 [nrows + 1]  offsets into col / val, rowptr[0] = 0
//   col      int32 [nnz]        global column ids (padded to 8 bytes)
//   val      fp32  [nnz]
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "pgcn_internal.h"

namespace {
const int64_t kMagic = 0x3130305253434750LL;   // "PGCSR001" little endian

struct File {
    FILE *f = nullptr;
    ~File() { if (f) fclose(f); }
};

int fail_io(const char *what, const char *path) {
    char buf[400];
    snprintf(buf, sizeof(buf), "%s %s: %s", what, path, strerror(errno));
    return pgcn_set_error(PGCN_EINVAL, buf);
}

bool put(FILE *f, const void *p, size_t bytes) { return bytes == 0 || fwrite(p, 1, bytes, f) == bytes; }
bool get(FILE *f, void *p, size_t bytes) { return bytes == 0 || fread(p, 1, bytes, f) == bytes; }
}  // namespace

extern "C" int pgcn_shard_write(const char *path, int64_t n_global, int32_t rank, int32_t nparts, int64_t nrows,
                                const int64_t *rows, const int64_t *rowptr, const int32_t *col, const float *val) {
    if (!path || n_global < 0 || nrows < 0 || rank < 0 || nparts <= rank || !rowptr || (nrows > 0 && !rows))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_shard_write: bad argument");
    if (n_global > 0x7fffffffLL) return pgcn_set_error(PGCN_EUNSUPPORTED, "pgcn_shard_write: column ids are 32-bit (n < 2^31)");
    const int64_t nnz = rowptr[nrows];
    if (rowptr[0] != 0 || nnz < 0 || (nnz > 0 && (!col || !val)))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_shard_write: bad row pointers");
    for (int64_t i = 0; i < nrows; ++i) {
        if (rowptr[i + 1] < rowptr[i]) return pgcn_set_error(PGCN_EINVAL, "pgcn_shard_write: row pointers not monotone");
        if (rows[i] < 0 || rows[i] >= n_global || (i > 0 && rows[i] <= rows[i - 1]))
            return pgcn_set_error(PGCN_EINVAL, "pgcn_shard_write: row ids must be ascending and inside the matrix");
    }
    File fh;
    fh.f = fopen(path, "wb");
    if (!fh.f) return fail_io("pgcn_shard_write: cannot create", path);
    const int64_t head[8] = {kMagic, n_global, nrows, nnz, rank, nparts, 0, 0};
    const int64_t zero = 0;
    bool ok = put(fh.f, head, sizeof(head)) && put(fh.f, rows, (size_t)nrows * 8) &&
              put(fh.f, rowptr, (size_t)(nrows + 1) * 8) && put(fh.f, col, (size_t)nnz * 4) &&
              put(fh.f, &zero, (size_t)((nnz & 1) * 4)) && put(fh.f, val, (size_t)nnz * 4);
    if (!ok) return fail_io("pgcn_shard_write: short write to", path);
    FILE *f = fh.f;
    fh.f = nullptr;                      // a failed flush at close is a failed write
    if (fclose(f) != 0) return fail_io("pgcn_shard_write: cannot flush", path);
    return PGCN_OK;
}

extern "C" int pgcn_shard_info(const char *path, int64_t out[6]) {
    if (!path || !out) return pgcn_set_error(PGCN_EINVAL, "pgcn_shard_info: null argument");
    File fh;
    fh.f = fopen(path, "rb");
    if (!fh.f) return fail_io("pgcn_shard_info: cannot open", path);
    int64_t head[8];
    if (!get(fh.f, head, sizeof(head)) || head[0] != kMagic)
        return pgcn_set_error2(PGCN_EINVAL, "pgcn_shard_info: not a PGCSR001 shard", path);
    if (head[1] < 0 || head[2] < 0 || head[3] < 0 || head[4] < 0 || head[5] <= head[4])
        return pgcn_set_error2(PGCN_EINVAL, "pgcn_shard_info: corrupt header", path);
    out[0] = head[1]; out[1] = head[2]; out[2] = head[3]; out[3] = head[4]; out[4] = head[5]; out[5] = head[6];
    return PGCN_OK;
}

extern "C" int pgcn_shard_read(const char *path, int64_t cap_rows, int64_t cap_nnz, int64_t *rows, int64_t *rowptr,
                               int32_t *col, float *val) {
    if (!path || !rows || !rowptr || (cap_nnz > 0 && (!col || !val)))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_shard_read: null argument");
    int64_t info[6];
    const int rc = pgcn_shard_info(path, info);
    if (rc != PGCN_OK) return rc;
    const int64_t nrows = info[1], nnz = info[2];
    if (cap_rows < nrows || cap_nnz < nnz) return pgcn_set_error(PGCN_ENOMEM, "pgcn_shard_read: capacity");
    File fh;
    fh.f = fopen(path, "rb");
    if (!fh.f) return fail_io("pgcn_shard_read: cannot open", path);
    int64_t head[8];
    int32_t pad;
    bool ok = get(fh.f, head, sizeof(head)) && get(fh.f, rows, (size_t)nrows * 8) &&
              get(fh.f, rowptr, (size_t)(nrows + 1) * 8) && get(fh.f, col, (size_t)nnz * 4) &&
              get(fh.f, &pad, (size_t)((nnz & 1) * 4)) && get(fh.f, val, (size_t)nnz * 4);
    if (!ok) return pgcn_set_error2(PGCN_EINVAL, "pgcn_shard_read: truncated shard", path);
    if (rowptr[0] != 0 || rowptr[nrows] != nnz) return pgcn_set_error2(PGCN_EINVAL, "pgcn_shard_read: corrupt row pointers", path);
    for (int64_t i = 0; i < nrows; ++i)
        if (rowptr[i + 1] < rowptr[i] || rows[i] < 0 || rows[i] >= info[0])
            return pgcn_set_error2(PGCN_EINVAL, "pgcn_shard_read: corrupt shard", path);
    for (int64_t k = 0; k < nnz; ++k)    // column ids become gather indices on the device
        if (col[k] < 0 || col[k] >= info[0])
            return pgcn_set_error2(PGCN_EINVAL, "pgcn_shard_read: column id outside the matrix", path);
    return PGCN_OK;
}
