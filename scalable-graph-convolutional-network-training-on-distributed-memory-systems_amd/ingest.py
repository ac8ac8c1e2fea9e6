"""Fast MatrixMarket ingest: multi-threaded C++ parser behind the C ABI.

``mmread(path)`` is a drop-in for ``scipy.io.mmread`` on the ``-a`` argument of
GPU/PGCN.py:171: it returns a ``scipy.sparse.coo_matrix`` with the same entries (fp32
values).  Coordinate real / integer / pattern files, general / symmetric /
skew-symmetric; anything else (array format, complex) is handed to scipy.
Host-side data plumbing: no GPU involved."""
from __future__ import annotations

import ctypes
import os

import numpy as np
import scipy.sparse as sp

from . import _lib


def mtx_info(path: str):
    L = _lib.lib()
    out = (ctypes.c_int64 * 4)()
    _lib.check(L.pgcn_mtx_info(os.fsencode(path), out), "pgcn_mtx_info")
    return {"nrows": out[0], "ncols": out[1], "stored": out[2], "pattern": bool(out[3] & 1),
            "symmetric": bool(out[3] & 2), "skew": bool(out[3] & 4), "integer": bool(out[3] & 8)}


def mmread(path: str, nthreads: int = 0) -> sp.coo_matrix:
    L = _lib.lib()
    out = (ctypes.c_int64 * 4)()
    rc = L.pgcn_mtx_info(os.fsencode(path), out)
    if rc == -5:                       # PGCN_EUNSUPPORTED: exotic flavour -> scipy
        from scipy.io import mmread as sp_mmread
        return sp.coo_matrix(sp_mmread(path))
    _lib.check(rc, "pgcn_mtx_info")
    stored = out[2]
    cap = stored * (2 if out[3] & 6 else 1)
    row = np.empty(max(cap, 1), dtype=np.int64)
    col = np.empty(max(cap, 1), dtype=np.int64)
    val = np.empty(max(cap, 1), dtype=np.float32)
    n = ctypes.c_int64()
    _lib.check(L.pgcn_mtx_read_coo(os.fsencode(path), cap, row.ctypes.data, col.ctypes.data,
                                   val.ctypes.data, ctypes.byref(n), nthreads), "pgcn_mtx_read_coo")
    k = n.value
    return sp.coo_matrix((val[:k], (row[:k], col[:k])), shape=(out[0], out[1]))


def comm_maps(row, col, partvec, rank: int, nranks: int, nthreads: int = 0):
    """compute_communication_maps (GPU/PGCN.py:37-51) on the native builder: returns
    (send_map, recv_map), dicts peer -> ascending int64 numpy array of GLOBAL ids, own rank
    absent.  ``row``/``col``: 0-based coordinates of the stored entries (every entry, or at
    least every entry with one end owned by ``rank``)."""
    L = _lib.lib()
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    pv = np.ascontiguousarray(partvec, dtype=np.int32)
    if row.shape != col.shape or row.ndim != 1:
        raise ValueError("row and col must be 1-D arrays of the same length")
    so = np.zeros(nranks + 1, dtype=np.int64)
    ro = np.zeros(nranks + 1, dtype=np.int64)
    args = (row.ctypes.data, col.ctypes.data, row.shape[0], pv.ctypes.data, pv.shape[0], rank, nranks,
            so.ctypes.data, ro.ctypes.data)
    _lib.check(L.pgcn_build_comm_maps(*args, None, 0, None, 0, nthreads), "pgcn_build_comm_maps")
    sid = np.empty(max(int(so[-1]), 1), dtype=np.int64)
    rid = np.empty(max(int(ro[-1]), 1), dtype=np.int64)
    _lib.check(L.pgcn_build_comm_maps(*args, sid.ctypes.data, sid.shape[0], rid.ctypes.data, rid.shape[0], nthreads),
               "pgcn_build_comm_maps")
    send = {q: sid[so[q]:so[q + 1]].copy() for q in range(nranks) if q != rank}
    recv = {q: rid[ro[q]:ro[q + 1]].copy() for q in range(nranks) if q != rank}
    return send, recv


def load_partition(path: str, partvec, rank: int, nthreads: int = 0) -> sp.coo_matrix:
    """mmread + get_partitiont_of_adjacency_matrix (GPU/PGCN.py:171,53-64): the rows of the
    matrix owned by ``rank`` as an n x n COO matrix (global coordinates, file order)."""
    L = _lib.lib()
    pv = np.ascontiguousarray(partvec, dtype=np.int32)
    info = mtx_info(path)
    n = ctypes.c_int64()
    a = (os.fsencode(path), pv.ctypes.data, pv.shape[0], rank)
    _lib.check(L.pgcn_load_mtx_partition(*a, 0, None, None, None, ctypes.byref(n), nthreads), "pgcn_load_mtx_partition")
    k = n.value
    row = np.empty(max(k, 1), dtype=np.int64)
    col = np.empty(max(k, 1), dtype=np.int64)
    val = np.empty(max(k, 1), dtype=np.float32)
    _lib.check(L.pgcn_load_mtx_partition(*a, k, row.ctypes.data, col.ctypes.data, val.ctypes.data, ctypes.byref(n),
                                         nthreads), "pgcn_load_mtx_partition")
    return sp.coo_matrix((val[:k], (row[:k], col[:k])), shape=(info["nrows"], info["ncols"]))


# --------------------------------------------------------------------------------------------
# Binary CSR shards (pgcn_shard_*): one file per rank with only that rank's rows.

SHARD_SUFFIX = ".pgcsr"


def shard_path(prefix: str, rank: int) -> str:
    return "%s.%d%s" % (prefix, rank, SHARD_SUFFIX)


def is_shard_prefix(path: str, rank: int = 0) -> bool:
    return os.path.exists(shard_path(path, rank))


def write_shard(path: str, n_global: int, rank: int, nparts: int, rows, rowptr, col, val) -> None:
    """One rank's row block: ``rows`` ascending global ids, CSR over them with GLOBAL int32 column ids."""
    L = _lib.lib()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    if rowptr.shape[0] != rows.shape[0] + 1 or col.shape[0] != val.shape[0] or (rowptr.size and rowptr[-1] != col.shape[0]):
        raise ValueError("inconsistent CSR arrays")
    _lib.check(L.pgcn_shard_write(os.fsencode(path), n_global, rank, nparts, rows.shape[0], rows.ctypes.data,
                                  rowptr.ctypes.data, col.ctypes.data, val.ctypes.data), "pgcn_shard_write")


def shard_info(path: str) -> dict:
    L = _lib.lib()
    out = (ctypes.c_int64 * 6)()
    _lib.check(L.pgcn_shard_info(os.fsencode(path), out), "pgcn_shard_info")
    return {"n": out[0], "nrows": out[1], "nnz": out[2], "rank": out[3], "nparts": out[4], "flags": out[5]}


def read_shard(path: str) -> dict:
    """{n, rank, nparts, rows int64, rowptr int64, col int32, val fp32} of one shard."""
    L = _lib.lib()
    info = shard_info(path)
    rows = np.empty(max(info["nrows"], 1), dtype=np.int64)
    rowptr = np.empty(info["nrows"] + 1, dtype=np.int64)
    col = np.empty(max(info["nnz"], 1), dtype=np.int32)
    val = np.empty(max(info["nnz"], 1), dtype=np.float32)
    _lib.check(L.pgcn_shard_read(os.fsencode(path), info["nrows"], info["nnz"], rows.ctypes.data, rowptr.ctypes.data,
                                 col.ctypes.data, val.ctypes.data), "pgcn_shard_read")
    info.update(rows=rows[:info["nrows"]], rowptr=rowptr, col=col[:info["nnz"]], val=val[:info["nnz"]])
    return info


def shard_coo(shard: dict):
    """(row, col, val) of a shard in GLOBAL coordinates (int64 rows and columns) -- what
    partition.build_partition_local takes."""
    counts = np.diff(shard["rowptr"])
    return np.repeat(shard["rows"], counts), shard["col"].astype(np.int64), shard["val"]


def write_shards(prefix: str, A, partvec, nparts: int) -> list:
    """Split a (scipy) matrix into ``nparts`` shards by the part vector; returns the paths."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    part = np.asarray(partvec, dtype=np.int64)
    paths = []
    for p in range(nparts):
        own = np.nonzero(part == p)[0]
        sub = A[own]
        path = shard_path(prefix, p)
        write_shard(path, A.shape[0], p, nparts, own, sub.indptr, sub.indices, sub.data)
        paths.append(path)
    return paths
