// pgcn_core.cpp -- error slot, ABI version, device query, host-side SpMM plan builder.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

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

// Plan: rows with <= chunk entries become one task; longer rows are cut into
// ceil(len/chunk) segments that write partial sums to consecutive slots and get
// one fix-up record.  Empty rows still get a (zero-length) task so that C is
// defined for them (C = 0, or C unchanged when accumulating).
extern "C" int pgcn_spmm_plan_host(const int64_t *rowptr, int64_t nrows, int32_t chunk,
                                   int32_t *tasks, int64_t cap_tasks, int32_t *fix,
                                   int64_t cap_fix, int64_t *ntasks, int64_t *nfix,
                                   int64_t *nslots) {
    if (!rowptr || nrows < 0 || chunk <= 0 || !ntasks || !nfix || !nslots)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: bad argument");
    if (nrows > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: nrows >= 2^31");
    int64_t nt = 0, nf = 0, ns = 0;
    for (int64_t r = 0; r < nrows; ++r) {
        const int64_t len = rowptr[r + 1] - rowptr[r];
        if (len < 0 || len > 0x7fffffffLL)
            return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: row pointer not monotone / row too long");
        if (len <= chunk) {
            if (tasks) {
                if (nt >= cap_tasks) return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_plan_host: tasks capacity");
                int32_t *t = tasks + 4 * nt;
                t[0] = (int32_t)r; t[1] = 0; t[2] = (int32_t)len; t[3] = -1;
            }
            ++nt;
        } else {
            const int64_t nseg = (len + chunk - 1) / chunk;
            // balance the segments of one row (all within one entry of each other)
            const int64_t seg = (len + nseg - 1) / nseg;
            if (ns + nseg > 0x7fffffffLL) return pgcn_set_error(PGCN_EINVAL, "pgcn_spmm_plan_host: too many slots");
            if (fix) {
                if (nf >= cap_fix) return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_plan_host: fix capacity");
                int32_t *x = fix + 4 * nf;
                x[0] = (int32_t)r; x[1] = (int32_t)ns; x[2] = (int32_t)nseg; x[3] = 0;
            }
            for (int64_t s = 0; s < nseg; ++s) {
                const int64_t off = s * seg;
                const int64_t l = (off + seg <= len) ? seg : (len - off);
                if (tasks) {
                    if (nt >= cap_tasks) return pgcn_set_error(PGCN_ENOMEM, "pgcn_spmm_plan_host: tasks capacity");
                    int32_t *t = tasks + 4 * nt;
                    t[0] = (int32_t)r; t[1] = (int32_t)off; t[2] = (int32_t)(l > 0 ? l : 0); t[3] = (int32_t)(ns + s);
                }
                ++nt;
            }
            ns += nseg;
            ++nf;
        }
    }
    *ntasks = nt; *nfix = nf; *nslots = ns;
    return PGCN_OK;
}
