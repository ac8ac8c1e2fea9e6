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
    assert len(names) >= 15
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), "libpgcn_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "%s is not bound in _lib.SIGNATURES" % n
    assert sorted(_lib.SIGNATURES) == names
    assert L.pgcn_abi_version() == 3


def _decode(tasks):
    kbeg = (tasks[:, 1].astype(np.int64) << 32) | (tasks[:, 0].astype(np.int64) & 0xffffffff)
    return kbeg, tasks[:, 2].astype(np.int64), tasks[:, 3].astype(np.int64)


def _check_plan(rowptr, tasks, fix, nslots, seg, chunk, cnt=None, small_row=0, ngroups=1):
    """Every stored entry is covered exactly once, by a task of its own row; slots of a row
    are consecutive; segments are sorted longest-first; sliced tasks stay inside a slice."""
    kbeg, ln, dst = _decode(tasks)
    nrows = rowptr.shape[0] - 1
    seg = list(seg)
    assert seg[0] == 0 and seg[-1] == tasks.shape[0]
    grp_of = np.zeros(tasks.shape[0], np.int64)
    assert (ln <= chunk).all()
    cover = np.zeros(rowptr[-1], np.int32)
    slot_row = {}
    for r, s, n in fix:
        for k in range(s, s + n):
            assert k not in slot_row
            slot_row[k] = r
    assert sorted(slot_row) == list(range(nslots))
    direct_rows = []
    for i in range(tasks.shape[0]):
        row = slot_row[dst[i]] if dst[i] >= 0 else ~dst[i]
        if dst[i] < 0:
            direct_rows.append(row)
        assert rowptr[row] <= kbeg[i] and kbeg[i] + ln[i] <= rowptr[row + 1]
        cover[kbeg[i]:kbeg[i] + ln[i]] += 1
        if cnt is not None and cnt[row].sum() > small_row and ln[i]:
            bounds = rowptr[row] + np.concatenate([[0], np.cumsum(cnt[row])])
            v = np.searchsorted(bounds, kbeg[i], side="right") - 1
            assert kbeg[i] + ln[i] <= bounds[v + 1]              # never crosses its (slice, group)
            s = v // ngroups
            grp_of[i] = v % ngroups
            assert seg[s] <= i < seg[s + 1]                      # and sits in that slice's segment
    for a, b in zip(seg, seg[1:]):                               # group-major, longest first inside a group
        g, l = grp_of[a:b], ln[a:b]
        assert (np.diff(g) >= 0).all()
        assert (np.diff(l)[np.diff(g) == 0] <= 0).all()
    assert (cover == 1).all()
    assert sorted(direct_rows + [r for r, _, _ in fix]) == list(range(nrows))   # each row written once


def test_plan_host():
    kernels = pkg("kernels")
    rowptr = np.array([0, 3, 3, 5000, 5001, 9000], dtype=np.int64)
    tasks, fix, nslots, seg = kernels.build_plan(rowptr, 1024)
    # rows 2 (4997 entries) and 4 (3999) are split into 5 and 4 balanced segments
    assert nslots == 9 and fix[:, :3].tolist() == [[2, 0, 5], [4, 5, 4]] and list(seg) == [0, 12]
    _check_plan(rowptr, tasks, fix[:, :3], nslots, seg, 1024)
    t2, f2, s2, g2 = kernels.build_plan(np.array([0, 1, 5], dtype=np.int64), 1024)
    assert t2 is None and f2 is None and s2 == 0
    t3, f3, s3, g3 = kernels.build_plan(np.array([0, 1, 5], dtype=np.int64), 1024, force=True)
    assert t3.tolist() == [[1, 0, 4, -2], [0, 0, 1, -1]] and s3 == 0
    # offsets beyond 2^31 survive the lo/hi split
    big = np.array([3_000_000_000, 3_000_000_700, 3_000_000_701], dtype=np.int64)
    t4, f4, s4, g4 = kernels.build_plan(big, 1024, force=True)
    kb, ln, dst = _decode(t4)
    assert kb.tolist() == [3_000_000_000, 3_000_000_700] and ln.tolist() == [700, 1]


def test_plan_host_sliced():
    kernels = pkg("kernels")
    rng = np.random.default_rng(0)
    S, nrows = 8, 300
    cnt = rng.integers(0, 60, (nrows, S)).astype(np.int32)
    cnt[5] = 0                               # an empty row
    cnt[6] = 0; cnt[6, 3] = 150              # a single-slice row -> direct write
    cnt[7, 2] = 700                          # a slice that needs chunking
    cnt[8] = 0; cnt[8, 1] = 3; cnt[8, 6] = 2  # a small row -> one unsliced task
    rowptr = np.concatenate([[0], np.cumsum(cnt.sum(1))]).astype(np.int64)
    for small in (0, 96):
        tasks, fix, nslots, seg = kernels.build_plan(rowptr, 256, cnt, small_row=small)
        _check_plan(rowptr, tasks, fix[:, :3], nslots, seg, 256, cnt, small)
        # the same counts read as 4 slices x 2 column groups
        t2, f2, n2, seg2 = kernels.build_plan(rowptr, 256, cnt, small_row=small, ngroups=2, group_min_row=0)
        assert len(list(seg2)) == 5
        _check_plan(rowptr, t2, f2[:, :3], n2, seg2, 256, cnt, small, ngroups=2)
        # with a large group_min_row every row is cut per slice only (groups merged)
        t3, f3, n3, seg3 = kernels.build_plan(rowptr, 256, cnt, small_row=small, ngroups=2, group_min_row=10**6)
        merged = cnt.reshape(nrows, 4, 2).sum(2)
        _check_plan(rowptr, t3, f3[:, :3], n3, seg3, 256, merged, small)
        assert t3.shape[0] < t2.shape[0]
        kbeg, ln, dst = _decode(tasks)
        assert ((dst == ~5) & (ln == 0)).sum() == 1
        assert ((dst == ~6) & (ln == 150)).sum() == 1
        if small:
            assert ((dst == ~8) & (ln == 5)).sum() == 1
        else:
            assert (dst == ~8).sum() == 0 and 8 in fix[:, 0].tolist()


def test_error_reporting_without_gpu():
    _lib = pkg("_lib")
    L = _lib.lib()
    nt = ctypes.c_int64()
    rc = L.pgcn_spmm_plan_host(None, None, None, 4, 1, 1, 0, 1024, 0, None, 0, None, 0, None, ctypes.byref(nt),
                               ctypes.byref(nt), ctypes.byref(nt))
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
