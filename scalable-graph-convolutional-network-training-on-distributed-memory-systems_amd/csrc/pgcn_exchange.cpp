// pgcn_exchange.cpp -- boundary-row exchange and weight-gradient all-reduce over RCCL.
//
// Replaces the per-peer blocking dist.send / dist.recv pairs of
// /root/reference/GPU/PGCN.py:99-115 (2.(P-1) host-ordered NCCL calls per
// exchange) == the MPI_Irecv / MPI_Isend / MPI_Waitany loop of
// Parallel-GCN/main.c:238-299, by ONE grouped all-to-all-v: inside a
// ncclGroupStart/End pair every peer gets one ncclSend and one ncclRecv, so
// RCCL drives all 7 xGMI links of an MI355X node concurrently (the mesh is
// point-to-point: each pairwise message has its own link).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include "pgcn_internal.h"

#define PGCN_NCCL_CHECK(expr)                                                         \
    do {                                                                              \
        ncclResult_t _r = (expr);                                                     \
        if (_r != ncclSuccess) return pgcn_set_error2(PGCN_ERCCL, #expr, ncclGetErrorString(_r)); \
    } while (0)

namespace {
struct Comm {
    ncclComm_t comm;
    int nranks;
    int rank;
};
}  // namespace

static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the C ABI");

extern "C" int pgcn_comm_unique_id(void *id128) {
    if (!id128) return pgcn_set_error(PGCN_EINVAL, "pgcn_comm_unique_id: null");
    ncclUniqueId id;
    PGCN_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return PGCN_OK;
}

extern "C" int pgcn_comm_init(void **comm, const void *id128, int32_t nranks, int32_t rank) {
    if (!comm || !id128 || nranks <= 0 || rank < 0 || rank >= nranks)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_comm_init: bad argument");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm *c = new Comm{nullptr, nranks, rank};
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return pgcn_set_error2(PGCN_ERCCL, "ncclCommInitRank", ncclGetErrorString(r));
    }
    *comm = c;
    return PGCN_OK;
}

extern "C" int pgcn_comm_destroy(void *comm) {
    if (!comm) return PGCN_OK;
    Comm *c = static_cast<Comm *>(comm);
    ncclResult_t r = ncclCommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return pgcn_set_error2(PGCN_ERCCL, "ncclCommDestroy", ncclGetErrorString(r));
    return PGCN_OK;
}

extern "C" int pgcn_exchange_alltoallv_f32(void *comm, const float *send, const int64_t *send_off,
                                           float *recv, const int64_t *recv_off, int32_t f,
                                           pgcn_stream_t stream) {
    if (!comm || !send_off || !recv_off || f <= 0)
        return pgcn_set_error(PGCN_EINVAL, "pgcn_exchange_alltoallv_f32: bad argument");
    Comm *c = static_cast<Comm *>(comm);
    hipStream_t s = (hipStream_t)stream;
    for (int q = 0; q < c->nranks; ++q) {
        if (send_off[q + 1] < send_off[q] || recv_off[q + 1] < recv_off[q])
            return pgcn_set_error(PGCN_EINVAL, "pgcn_exchange_alltoallv_f32: offsets not monotone");
    }
    if (send_off[c->rank + 1] != send_off[c->rank] || recv_off[c->rank + 1] != recv_off[c->rank])
        return pgcn_set_error(PGCN_EINVAL, "pgcn_exchange_alltoallv_f32: own-rank segment must be empty");
    // A group that was opened is ALWAYS closed: returning between ncclGroupStart and ncclGroupEnd would leave the communicator
    // inside an open group and the next collective of this rank (and with it every peer) would hang.  The first failure is
    // remembered, the remaining calls of the group are skipped, the group is ended, and the first failure is reported.
    PGCN_NCCL_CHECK(ncclGroupStart());
    ncclResult_t first = ncclSuccess;
    const char *what = nullptr;
    bool null_slab = false;
    for (int q = 0; q < c->nranks && first == ncclSuccess && !null_slab; ++q) {
        if (q == c->rank) continue;
        const int64_t ns = send_off[q + 1] - send_off[q];
        const int64_t nr = recv_off[q + 1] - recv_off[q];
        if ((ns > 0 && !send) || (nr > 0 && !recv)) { null_slab = true; break; }
        if (ns > 0) {
            first = ncclSend(send + send_off[q] * f, (size_t)(ns * f), ncclFloat, q, c->comm, s);
            if (first != ncclSuccess) { what = "ncclSend"; break; }
        }
        if (nr > 0) {
            first = ncclRecv(recv + recv_off[q] * f, (size_t)(nr * f), ncclFloat, q, c->comm, s);
            if (first != ncclSuccess) { what = "ncclRecv"; break; }
        }
    }
    const ncclResult_t end = ncclGroupEnd();
    if (null_slab) return pgcn_set_error(PGCN_EINVAL, "pgcn_exchange_alltoallv_f32: null send / recv slab with a non-empty segment");
    if (first != ncclSuccess) return pgcn_set_error2(PGCN_ERCCL, what, ncclGetErrorString(first));
    if (end != ncclSuccess) return pgcn_set_error2(PGCN_ERCCL, "ncclGroupEnd", ncclGetErrorString(end));
    return PGCN_OK;
}

extern "C" int pgcn_allreduce_sum_f32(void *comm, float *buf, int64_t count, pgcn_stream_t stream) {
    if (!comm || count < 0) return pgcn_set_error(PGCN_EINVAL, "pgcn_allreduce_sum_f32: bad argument");
    if (count == 0) return PGCN_OK;
    if (!buf) return pgcn_set_error(PGCN_EINVAL, "pgcn_allreduce_sum_f32: null buffer");
    Comm *c = static_cast<Comm *>(comm);
    PGCN_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, (hipStream_t)stream));
    return PGCN_OK;
}
