// pgcn_mtx.cpp -- parallel MatrixMarket (coordinate) reader, host only.
//
// Replaces scipy.io.mmread on the -a input of /root/reference/GPU/PGCN.py:171 (every rank
// parses the whole text matrix; scipy needs minutes at Reddit scale).  The file is mmap'ed,
// cut into per-thread chunks at line boundaries, lines are counted, then parsed in place with
// a hand-rolled integer / decimal scanner (falls back to strtod for anything unusual) into
// caller-provided arrays.  Semantics follow scipy's mmread for coordinate files:
//   real | integer | pattern (value 1)  x  general | symmetric | skew-symmetric
//   1-based indices -> 0-based; symmetric files are expanded (mirror of every off-diagonal
//   entry appended after the stored entries, negated for skew-symmetric).
// complex / hermitian / array files are reported as PGCN_EUNSUPPORTED.
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "pgcn_internal.h"

namespace {

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p) munmap(const_cast<char *>(p), n);
        if (fd >= 0) close(fd);
    }
};

bool map_file(const char *path, Mapped &m) {
    m.fd = open(path, O_RDONLY);
    if (m.fd < 0) return false;
    struct stat st;
    if (fstat(m.fd, &st) != 0) return false;
    m.n = (size_t)st.st_size;
    if (m.n == 0) return true;
    void *a = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
    if (a == MAP_FAILED) return false;
    m.p = static_cast<const char *>(a);
    madvise(a, m.n, MADV_SEQUENTIAL);
    return true;
}

struct Header {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    bool pattern = false, integer = false, symmetric = false, skew = false;
    size_t body = 0;   // offset of the first entry line
};

inline const char *line_end(const char *p, const char *end) {
    const char *q = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    return q ? q : end;
}

int parse_header(const Mapped &m, Header &h) {
    const char *p = m.p, *end = m.p + m.n;
    if (!p || m.n < 14) return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx: empty file");
    const char *e = line_end(p, end);
    std::string banner(p, e);
    std::transform(banner.begin(), banner.end(), banner.begin(), ::tolower);
    if (banner.compare(0, 14, "%%matrixmarket") != 0) return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx: missing %%MatrixMarket banner");
    if (banner.find("matrix") == std::string::npos || banner.find("coordinate") == std::string::npos)
        return pgcn_set_error(PGCN_EUNSUPPORTED, "pgcn_mtx: only 'matrix coordinate' files are supported");
    if (banner.find("complex") != std::string::npos || banner.find("hermitian") != std::string::npos)
        return pgcn_set_error(PGCN_EUNSUPPORTED, "pgcn_mtx: complex / hermitian files are not supported");
    h.pattern = banner.find("pattern") != std::string::npos;
    h.integer = banner.find("integer") != std::string::npos;
    h.skew = banner.find("skew-symmetric") != std::string::npos;
    h.symmetric = !h.skew && banner.find("symmetric") != std::string::npos;
    p = (e < end) ? e + 1 : end;
    while (p < end) {   // comments and blank lines
        e = line_end(p, end);
        const char *q = p;
        while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
        if (q < e && *q != '%') break;
        p = (e < end) ? e + 1 : end;
    }
    if (p >= end) return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx: no size line");
    char *q = nullptr;
    h.nrows = strtoll(p, &q, 10);
    h.ncols = strtoll(q, &q, 10);
    h.nnz = strtoll(q, &q, 10);
    if (h.nrows < 0 || h.ncols < 0 || h.nnz < 0) return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx: bad size line");
    e = line_end(p, end);
    h.body = (size_t)(((e < end) ? e + 1 : end) - m.p);
    return PGCN_OK;
}

inline void skip_ws(const char *&p, const char *e) {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
}

inline bool parse_i64(const char *&p, const char *e, int64_t &v) {
    skip_ws(p, e);
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    if (p >= e || *p < '0' || *p > '9') return false;
    int64_t x = 0;
    while (p < e && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); ++p; }
    v = neg ? -x : x;
    return true;
}

// decimal -> double; exact for <= 15 significant digits and |exp10| <= 22 (the fast path every
// text matrix hits), strtod otherwise => same double, hence the same float, as scipy's float().
inline bool parse_f64(const char *&p, const char *e, double &v) {
    skip_ws(p, e);
    const char *s = p;
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    uint64_t mant = 0;
    int digits = 0, exp10 = 0;
    bool any = false;
    while (p < e && *p >= '0' && *p <= '9') {
        if (digits < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) ++digits; } else ++exp10;
        ++p; any = true;
    }
    if (p < e && *p == '.') {
        ++p;
        while (p < e && *p >= '0' && *p <= '9') {
            if (digits < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) ++digits; --exp10; }
            ++p; any = true;
        }
    }
    if (!any) return false;
    if (p < e && (*p == 'e' || *p == 'E' || *p == 'd' || *p == 'D')) {
        const char *q = p + 1;
        bool eneg = false;
        if (q < e && (*q == '-' || *q == '+')) { eneg = *q == '-'; ++q; }
        if (q < e && *q >= '0' && *q <= '9') {
            int ex = 0;
            while (q < e && *q >= '0' && *q <= '9') { if (ex < 10000) ex = ex * 10 + (*q - '0'); ++q; }
            exp10 += eneg ? -ex : ex;
            p = q;
        }
    }
    static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    if (digits <= 15 && exp10 >= -22 && exp10 <= 22) {
        double d = (double)mant;
        d = exp10 < 0 ? d / p10[-exp10] : d * p10[exp10];
        v = neg ? -d : d;
        return true;
    }
    std::string tmp(s, p);   // rare: long mantissa / huge exponent -> libc
    for (auto &ch : tmp) if (ch == 'd' || ch == 'D') ch = 'e';
    v = strtod(tmp.c_str(), nullptr);
    return true;
}

}  // namespace

extern "C" int pgcn_mtx_info(const char *path, int64_t out[4]) {
    if (!path || !out) return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx_info: null argument");
    Mapped m;
    if (!map_file(path, m)) return pgcn_set_error2(PGCN_EINVAL, "pgcn_mtx_info: cannot open", path);
    Header h;
    int rc = parse_header(m, h);
    if (rc != PGCN_OK) return rc;
    out[0] = h.nrows; out[1] = h.ncols; out[2] = h.nnz;
    out[3] = (h.pattern ? 1 : 0) | (h.symmetric ? 2 : 0) | (h.skew ? 4 : 0) | (h.integer ? 8 : 0);
    return PGCN_OK;
}

extern "C" int pgcn_mtx_read_coo(const char *path, int64_t cap, int64_t *row, int64_t *col, float *val,
                                 int64_t *nnz_out, int32_t nthreads) {
    if (!path || !row || !col || !val || !nnz_out || cap < 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx_read_coo: bad argument");
    Mapped m;
    if (!map_file(path, m)) return pgcn_set_error2(PGCN_EINVAL, "pgcn_mtx_read_coo: cannot open", path);
    Header h;
    int rc = parse_header(m, h);
    if (rc != PGCN_OK) return rc;
    if (h.nnz > cap) return pgcn_set_error(PGCN_ENOMEM, "pgcn_mtx_read_coo: capacity smaller than the stored entries");
    if (nthreads < 1) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    const char *b = m.p + h.body, *end = m.p + m.n;
    const size_t len = (size_t)(end - b);
    if (len < (size_t)nthreads * 4096) nthreads = 1;
    // chunk boundaries at line starts
    std::vector<const char *> cut(nthreads + 1);
    cut[0] = b; cut[nthreads] = end;
    for (int t = 1; t < nthreads; ++t) {
        const char *p = b + len * (size_t)t / (size_t)nthreads;
        p = line_end(p, end);
        cut[t] = (p < end) ? p + 1 : end;
    }
    // pass 1: entry lines per chunk
    std::vector<int64_t> cnt(nthreads, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t]() {
                int64_t c = 0;
                for (const char *p = cut[t]; p < cut[t + 1];) {
                    const char *e = line_end(p, cut[t + 1]);
                    const char *q = p;
                    skip_ws(q, e);
                    if (q < e && *q != '%') ++c;
                    p = (e < cut[t + 1]) ? e + 1 : cut[t + 1];
                }
                cnt[t] = c;
            });
        for (auto &x : th) x.join();
    }
    std::vector<int64_t> off(nthreads + 1, 0);
    for (int t = 0; t < nthreads; ++t) off[t + 1] = off[t] + cnt[t];
    if (off[nthreads] != h.nnz)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx_read_coo: number of entry lines differs from the size line");
    // pass 2: parse
    std::vector<int> bad(nthreads, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t]() {
                int64_t k = off[t];
                for (const char *p = cut[t]; p < cut[t + 1];) {
                    const char *e = line_end(p, cut[t + 1]);
                    const char *q = p;
                    skip_ws(q, e);
                    if (q < e && *q != '%') {
                        int64_t i, j;
                        double v = 1.0;
                        bool ok = parse_i64(q, e, i) && parse_i64(q, e, j);
                        if (ok && !h.pattern) ok = parse_f64(q, e, v);
                        if (!ok || i < 1 || j < 1 || i > h.nrows || j > h.ncols) { bad[t] = 1; break; }
                        row[k] = i - 1; col[k] = j - 1; val[k] = (float)v;
                        ++k;
                    }
                    p = (e < cut[t + 1]) ? e + 1 : cut[t + 1];
                }
            });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < nthreads; ++t)
        if (bad[t]) return pgcn_set_error(PGCN_EINVAL, "pgcn_mtx_read_coo: malformed entry line or index out of range");
    int64_t n = h.nnz;
    if (h.symmetric || h.skew) {   // append the mirror of every off-diagonal entry
        int64_t extra = 0;
        for (int64_t k = 0; k < h.nnz; ++k) extra += row[k] != col[k];
        if (n + extra > cap) return pgcn_set_error(PGCN_ENOMEM, "pgcn_mtx_read_coo: capacity smaller than the expanded symmetric matrix");
        for (int64_t k = 0; k < h.nnz; ++k)
            if (row[k] != col[k]) { row[n] = col[k]; col[n] = row[k]; val[n] = h.skew ? -val[k] : val[k]; ++n; }
    }
    *nnz_out = n;
    return PGCN_OK;
}
