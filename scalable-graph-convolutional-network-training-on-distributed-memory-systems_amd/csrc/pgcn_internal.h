// pgcn_internal.h -- shared by the translation units of libpgcn_hip.so (not installed).
#ifndef PGCN_INTERNAL_H
#define PGCN_INTERNAL_H

#include "../../include/pgcn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
// records `msg` (plus optional detail) in the calling thread's error slot, returns `code`
int pgcn_set_error(int code, const char *msg);
int pgcn_set_error2(int code, const char *msg, const char *detail);
#ifdef __cplusplus
}
#endif

#define PGCN_HIP_CHECK(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) return pgcn_set_error2(PGCN_EHIP, #expr, hipGetErrorString(_e)); \
    } while (0)

#endif
