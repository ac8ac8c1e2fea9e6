// pgcn_once.h -- first-call set-up of a launch site (included by the .hip files that raise a kernel's dynamic LDS limit).
#ifndef PGCN_ONCE_H
#define PGCN_ONCE_H
#include <hip/hip_runtime.h>

#include "pgcn_internal.h"

#include <mutex>
// First-call set-up of a launch site, once per device (hipFuncSetAttribute is per device), safe against two host threads
// making their first call together: PSpMM.backward runs on PyTorch's autograd thread while the main thread may be inside
// a forward on another stream.  `fn` returns PGCN_OK or an error code (nothing is marked then).
struct PgcnPerDeviceOnce {
    std::mutex mu;
    bool done[64] = {};
    template <class F>
    int run(F &&fn) {
        int dev = 0;
        PGCN_HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (dev >= 0 && dev < 64 && done[dev]) return PGCN_OK;
        const int rc = fn();
        if (rc != PGCN_OK) return rc;
        if (dev >= 0 && dev < 64) done[dev] = true;
        return PGCN_OK;
    }
};
#endif
