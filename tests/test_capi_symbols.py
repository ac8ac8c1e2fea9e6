"""The C-ABI library loads on a CPU-only box and exports every symbol include/pgcn_hip.h
declares (no compute calls here); host-only entry points behave."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pkg


def _declared():
    src = open(os.path.join(ROOT, "include", "pgcn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pgcn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    _lib = pkg("_lib")
    names = _declared()
    assert len(names) >= 13
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), "libpgcn_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "%s is not bound in _lib.SIGNATURES" % n
    assert sorted(_lib.SIGNATURES) == names
    assert L.pgcn_abi_version() == 1


def test_plan_host():
    kernels = pkg("kernels")
    rowptr = np.array([0, 3, 3, 5000, 5001, 9000], dtype=np.int64)
    tasks, fix, nslots = kernels.build_plan(rowptr, 1024)
    # rows 2 (4997 entries) and 4 (3999) are split into 5 and 4 balanced segments
    assert nslots == 9 and fix.tolist() == [[2, 0, 5, 0], [4, 5, 4, 0]]
    for r in range(5):
        t = tasks[tasks[:, 0] == r]
        assert t[:, 2].sum() == rowptr[r + 1] - rowptr[r]
        assert (t[:, 2] <= 1024).all()
        off = 0
        for row in t:                      # contiguous cover of the row
            assert row[1] == off
            off += row[2]
    assert sorted(tasks[tasks[:, 3] >= 0][:, 3].tolist()) == list(range(9))
    t2, f2, s2 = kernels.build_plan(np.array([0, 1, 5], dtype=np.int64), 1024)
    assert t2 is None and f2 is None and s2 == 0


def test_error_reporting_without_gpu():
    _lib = pkg("_lib")
    L = _lib.lib()
    nt = ctypes.c_int64()
    rc = L.pgcn_spmm_plan_host(None, 4, 1024, None, 0, None, 0, ctypes.byref(nt), ctypes.byref(nt),
                               ctypes.byref(nt))
    assert rc == -1 and b"pgcn_spmm_plan_host" in L.pgcn_last_error()
    with pytest.raises(_lib.PgcnError):
        _lib.check(rc, "plan")
    # bad sizes are rejected before any HIP call is made
    assert L.pgcn_spmm_csr_f32(None, None, None, 5, None, 4, None, 4, 8, 0, None) == -1
    assert L.pgcn_gather_rows_f32(None, 8, None, 3, None, 8, 8, None) == -1


def test_product_refuses_to_run_without_a_hip_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    kernels, _lib = pkg("kernels"), pkg("_lib")
    with pytest.raises(_lib.PgcnError):
        kernels.HipKernels(torch.device("cuda:0"))
    P = pkg("PGCN")
    P._kernel_provider = None
    P.device = torch.device("cpu")
    with pytest.raises(_lib.PgcnError):
        P._provider()
