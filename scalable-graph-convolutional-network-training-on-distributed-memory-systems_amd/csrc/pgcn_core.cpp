// pgcn_core.cpp -- error slot, ABI version, device query, host-side SpMM plan builder.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "pgcn_internal.h"

namespace {
thread_local char g_err[512] = "";
}

extern "C" int pgcn_set_error(int code, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

extern "C" int pgcn_set_error2(int code, const char *msg, const char *detail) {
    snprintf(g_err, sizeof(g_err), "%s: %s", msg ? msg : "", detail ? detail : "");
    return code;
}

extern "C" const char *pgcn_last_error(void) { return g_err; }

extern "C" int pgcn_abi_version(void) { return PGCN_ABI_VERSION; }

extern "C" int pgcn_device_info(int32_t device, int64_t out[4]) {
    if (!out) return pgcn_set_error(PGCN_EINVAL, "pgcn_device_info: null out");
    hipDeviceProp_t p;
    PGCN_HIP_CHECK(hipGetDeviceProperties(&p, device));
    out[0] = p.multiProcessorCount;
    out[1] = p.warpSize;
    int arch = 0;
    // gcnArchName looks like "gfx950:sramecc+:xnack-"
    if (strncmp(p.gcnArchName, "gfx", 3) == 0) sscanf(p.gcnArchName + 3, "%d", &arch);
    out[2] = arch;
    out[3] = p.l2CacheSize;
    return PGCN_OK;
}

// Plan.  Work is cut into TASKS = {kbeg lo, kbeg hi, length, dst}: `length` stored
// entries starting at absolute offset kbeg; dst >= 0 is a partial-sum slot, dst < 0
// means "write row ~dst of C directly".
//   * nslices > 1 ("XCD-sliced"): the entries of every row are stored grouped by
//     slice = col % nslices (slice_cnt[r*nslices + s] entries each); a task never
//     crosses a slice, and the task list is grouped BY SLICE (seg[s]..seg[s+1]) so that
//     the kernel can hand slice s to the workgroups that run on XCD s.  Each XCD's
//     private L2 then only ever sees 1/nslices of the rows of B.
//   * rows with at most `small_row` entries are NOT sliced: one task, direct write,
//     placed round-robin over the segments (slicing them would multiply the per-task
//     latency chain and the partial-sum traffic for a handful of entries).
//   * ngroups > 1 ("column groups"): inside a slice the entries are further grouped by
//     column range g (slice_cnt has nslices*ngroups columns, index s*ngroups + g); a task
//     never crosses a group and every segment is ordered group-major, so at any time the
//     tasks resident on an XCD read one column group: (XCD, time) slicing makes the L2
//     working set 1/(8*ngroups) of B.
//   * pieces longer than `chunk` are cut into balanced segments.
//   * a row with exactly one task writes C directly; a row with several tasks gets
//     consecutive partial-sum slots and one fix-up record {row, first slot, #tasks},
//     combined in slot order by the fix-up kernel (deterministic, no atomics).
//   * empty rows still get one zero-length task so that C is defined for them.
//   * inside every segment the tasks are ordered by decreasing length (stable):
//     longest-first scheduling, and the tasks sharing a wavefront have equal trip counts.
namespace {
struct TaskRec { int64_t kbeg; int32_t len; int32_t dst; int32_t grp; };
}

extern "C" int pgcn_spmm_plan_host(const int64_t *rowptr, const int32_t *slice_cnt,
                                   const uint8_t *row_flags, int64_t nrows, int32_t nslices,
                                   int32_t ngroups, int32_t group_min_row, int32_t chunk,
                                   int32_t small_row,
                                   int32_t *tasks, int64_t cap_tasks, int32_t *fix,
                                   int64_t cap_fix, int64_t *seg, int64_t *ntasks,
                                   int64_t *nfix, int64_t *nslots) {
    return pgcn_spmm_plan_host_ex(rowptr, slice_cnt, row_flags, nrows, nslices, ngroups, group_min_row, chunk, small_row, 0, tasks,
                                  cap_tasks, fix, cap_fix, seg, ntasks, nfix, nslots);
}

//   * plan_flags & PGCN_PLAN_AFFINE_SMALL (r05): an unsliced short row's task runs on the segment (XCD) of its fullest slice.
//     (r05 also measured rows of small_row < entries <= pair_row cut per PAIR of adjacent slices: slower on all three benchmark
//     graphs, removed in r06 -- profiles/r05_pair_rows.txt, HISTORY.md section 10.)
extern "C" int pgcn_spmm_plan_host_ex(const int64_t *rowptr, const int32_t *slice_cnt,
                                      const uint8_t *row_flags, int64_t nrows, int32_t nslices,
                                      int32_t ngroups, int32_t group_min_row, int32_t chunk,
                                      int32_t small_row, int32_t plan_flags,
                                      int32_t *tasks, int64_t cap_tasks, int32_t *fix,
                                      int64_t cap_fix, int64_t *seg, int64_t *ntasks,
                                      int64_t *nfix, int64_t *nslots) {
    if (!rowptr || nrows < 0 || chunk <= 0 || small_row < 0 || !ntasks || !nfix || !nslots || !seg ||
        nslices < 1 || nslices > PGCN_MAX_SLICES || ngroups < 1 || ngroups > PGCN_MAX_COL_GROUPS ||
        (nslices * ngroups > 1 && !slice_cnt))
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: bad argument");
    if (nrows > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: nrows >= 2^31");
    const int S = nslices;           // XCD slices = task segments
    const int G = ngroups;           // column groups inside a slice (time-ordered)
    const int V = S * G;             // "virtual slices": entries of a row are grouped by v = s*G + g
    // rows shorter than group_min_row are cut per slice only: their column groups are merged
    // back (cutting a medium row 8*G ways would only multiply tiny tasks and partial sums)
    const bool affine = (plan_flags & PGCN_PLAN_AFFINE_SMALL) != 0 && V > 1;   // unsliced rows go to the segment of their fullest slice
    // the segment of an UNSLICED row: round-robin, or (affine) the XCD whose slice holds most of its entries -- those reads then
    // meet the rows of B that XCD's sliced tasks keep in its L2 (ties and empty rows: round-robin)
    auto small_seg = [&](int64_t r) -> int {
        if (!affine) return (int)(r % S);
        int best = (int)(r % S);
        int64_t most = 0;
        for (int g = 0; g < G; ++g) most += slice_cnt[r * V + best * G + g];
        for (int q = 0; q < S; ++q) {
            int64_t c = 0;
            for (int g = 0; g < G; ++g) c += slice_cnt[r * V + q * G + g];
            if (c > most) { most = c; best = q; }
        }
        return best;
    };
    // the segment (XCD) that runs piece v of row r
    auto seg_of = [&](int64_t, int64_t, int v) -> int { return v / G; };
    auto piece_len = [&](int64_t r, int64_t len, int v) -> int64_t {
        if (V == 1) return len;
        if (G == 1 || len >= group_min_row) return slice_cnt[r * V + v];
        if (v % G != 0) return 0;
        int64_t l = 0;
        for (int g = 0; g < G; ++g) l += slice_cnt[r * V + v + g];
        return l;
    };
    if (small_row > chunk) small_row = chunk;
    // pass 1: tasks per segment; slots per row
    int64_t per_slice[PGCN_MAX_SLICES] = {0};
    int64_t nt = 0, nf = 0, ns = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        const int64_t len = rowptr[r + 1] - rowptr[r];
        if (len < 0 || len > 0x7fffffffLL)
            return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: row pointer not monotone / row too long");
        if (V > 1) {
            int64_t sum = 0;
            for (int v = 0; v < V; ++v) {
                if (slice_cnt[r * V + v] < 0)
                    return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: negative slice count");
                sum += slice_cnt[r * V + v];
            }
            if (sum != len)
                return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: slice counts do not add up to the row length");
        }
        const bool shared = row_flags && row_flags[r];   // other kernels add into this row
        if (shared && len == 0) continue;                // nothing of ours to add
        if (len <= small_row || (V == 1 && len <= chunk)) {   // one unsliced task (also: empty row)
            per_slice[small_seg(r)] += 1;
            nt += 1;
            if (shared) { ns += 1; ++nf; }
            continue;
        }
        int64_t row_tasks = 0;
        for (int v = 0; v < V; ++v) {
            const int64_t l = piece_len(r, len, v);
            const int64_t k = (l + chunk - 1) / chunk;
            per_slice[seg_of(r, len, v)] += k;
            row_tasks += k;
        }
        nt += row_tasks;
        if (row_tasks > 1 || shared) { ns += row_tasks; ++nf; }
    }
    if (ns > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: too many slots");
    seg[0] = 0;
    for (int s = 0; s < S; ++s) seg[s + 1] = seg[s] + per_slice[s];
    *ntasks = nt; *nfix = nf; *nslots = ns;
    if (!tasks) return PGCN_OK;
    if (nt > cap_tasks) return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_plan_host: tasks capacity");
    if (nf > 0 && (!fix || nf > cap_fix)) return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_plan_host: fix capacity");
    // pass 2: emit into a scratch list, segment by segment
    TaskRec *rec = (TaskRec *)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(TaskRec));
    if (!rec) return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_plan_host: out of host memory");
    int64_t cur[PGCN_MAX_SLICES];
    for (int s = 0; s < S; ++s) cur[s] = seg[s];
    int64_t slot = 0, fi = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        const int64_t len = rowptr[r + 1] - rowptr[r];
        const bool shared = row_flags && row_flags[r];
        if (shared && len == 0) continue;
        if (len <= small_row || (V == 1 && len <= chunk)) {
            if (shared) {
                int32_t *x = fix + 4 * fi++;
                x[0] = (int32_t)r; x[1] = (int32_t)slot; x[2] = 1; x[3] = 0;
                rec[cur[small_seg(r)]++] = TaskRec{rowptr[r], (int32_t)len, (int32_t)slot++, 0};
            } else {
                rec[cur[small_seg(r)]++] = TaskRec{rowptr[r], (int32_t)len, ~(int32_t)r, 0};
            }
            continue;
        }
        int64_t row_tasks = 0;
        for (int v = 0; v < V; ++v) {
            const int64_t l = piece_len(r, len, v);
            row_tasks += (l + chunk - 1) / chunk;
        }
        const bool direct = row_tasks == 1 && !shared;
        if (!direct) {
            int32_t *x = fix + 4 * fi++;
            x[0] = (int32_t)r; x[1] = (int32_t)slot; x[2] = (int32_t)row_tasks; x[3] = 0;
        }
        int64_t off = 0;
        for (int v = 0; v < V; ++v) {
            const int64_t l = piece_len(r, len, v);
            const int64_t k = (l + chunk - 1) / chunk;
            if (k > 0) {
                const int64_t piece = (l + k - 1) / k;   // balanced segments
                for (int64_t j = 0; j < k; ++j) {
                    const int64_t o = j * piece;
                    const int64_t ll = (o + piece <= l) ? piece : (l - o);
                    rec[cur[seg_of(r, len, v)]++] = TaskRec{rowptr[r] + off + o, (int32_t)ll,
                                                            direct ? ~(int32_t)r : (int32_t)slot++, v % G};
                }
            }
            off += l;
        }
    }
    // inside each segment: column group by column group (the workgroups of an XCD sweep the
    // column space together, so their common working set is one group = a few MB of B), longest
    // first inside a group
    for (int s = 0; s < S; ++s)
        std::stable_sort(rec + seg[s], rec + seg[s + 1], [](const TaskRec &a, const TaskRec &b) {
            return a.grp != b.grp ? a.grp < b.grp : a.len > b.len;
        });
    for (int64_t i = 0; i < nt; ++i) {
        int32_t *t = tasks + 4 * i;
        t[0] = (int32_t)(uint32_t)((uint64_t)rec[i].kbeg & 0xffffffffu);
        t[1] = (int32_t)(uint32_t)((uint64_t)rec[i].kbeg >> 32);
        t[2] = rec[i].len;
        t[3] = rec[i].dst;
    }
    free(rec);
    return PGCN_OK;
}
