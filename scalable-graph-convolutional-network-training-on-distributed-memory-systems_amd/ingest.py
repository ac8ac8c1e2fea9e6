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
