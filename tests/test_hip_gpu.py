"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Needs an MI355X:  python -m pytest tests -m gpu

Tolerances: integer / index / copy work is bit-exact; fp32 SpMM agrees with the oracle
within 1e-5 relative to the largest magnitude (north_star: "within 1e-5 relative fp32";
the only difference is FMA contraction + the fixed segment-combine order of split rows)."""
import contextlib
import ctypes
import io
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.io import mmread

from conftest import (SPMM_CASES, SPMM_CASES_MORE, TRAIN_CASES, TRAIN_CASES_MORE, free_port, golden, golden_inputs, gpath,
                      held_to_fixture, pkg, read_partvec, rel_err)
from oracle import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    d = torch.device("cuda:0")
    torch.cuda.set_device(d)
    return d


@pytest.fixture(scope="module")
def K(dev):
    return pkg("kernels").HipKernels(dev)


def _host_csr(A):
    partition = pkg("partition")
    A = sp.csr_matrix(A)
    A.sort_indices()
    return partition.HostCSR(A.shape[0], A.shape[1], torch.from_numpy(A.indptr.astype(np.int64)),
                             torch.from_numpy(A.indices.astype(np.int32)),
                             torch.from_numpy(A.data.astype(np.float32)))


def _spmm_gpu(K, dev, A, B, chunk=None, accumulate_into=None, nslices=1):
    h = _host_csr(A) if nslices == 1 else pkg("partition").csr_from_scipy(A, nslices)
    if chunk is not None:
        old, K.chunk = K.chunk, chunk
    d = K.prepare(h)
    if chunk is not None:
        K.chunk = old
    Bd = torch.from_numpy(B).to(dev)
    if accumulate_into is None:
        C = torch.full((A.shape[0], B.shape[1]), float("nan"), device=dev)
        K.spmm(d, Bd, C)
    else:
        C = torch.from_numpy(accumulate_into).to(dev)
        K.spmm(d, Bd, C, accumulate=True)
    torch.cuda.synchronize()
    return C.cpu().numpy(), d


def test_library_and_device(K):
    info = K.device_info()
    assert info["gfx"] == 950 and info["wave"] == 64 and info["cus"] >= 200, info


@pytest.mark.parametrize("mtx", ["karate.mtx", "gemat11.mtx", "gemat11p.A.mtx"])
@pytest.mark.parametrize("f", [1, 2, 3, 4, 8, 16, 20, 64, 100, 128, 256, 300, 512])
@pytest.mark.parametrize("nslices", [1, 8])
def test_spmm_matches_oracle(K, dev, mtx, f, nslices):
    A = sp.csr_matrix(mmread(gpath(mtx))).astype(np.float32)
    rng = np.random.default_rng(f)
    B = rng.random((A.shape[1], f), dtype=np.float32) * 2 - 1
    got, d = _spmm_gpu(K, dev, A, B, nslices=nslices)
    assert d.nslices == nslices
    assert rel_err(got, oracle.spmm(A, B)) < TOL
    got_t, _ = _spmm_gpu(K, dev, sp.csr_matrix(A.T), B, nslices=nslices)   # backward operand (A^T)
    assert rel_err(got_t, oracle.spmm(sp.csr_matrix(A.T), B)) < TOL


def test_spmm_unaligned_leading_dimension(K, dev):
    """f % 4 == 0 but a padded / offset view forces the scalar kernel shape."""
    A = sp.csr_matrix(mmread(gpath("gemat11p.A.mtx"))).astype(np.float32)
    rng = np.random.default_rng(1)
    Bp = torch.from_numpy(rng.random((A.shape[1], 19), dtype=np.float32)).to(dev)
    B = Bp[:, 1:17]                                               # ld 19, base offset 4 bytes
    d = K.prepare(_host_csr(A))
    C = torch.empty((A.shape[0], 16), device=dev)
    K.spmm(d, B, C)
    torch.cuda.synchronize()
    assert rel_err(C.cpu().numpy(), oracle.spmm(A, B.cpu().numpy().copy())) < TOL


@pytest.mark.parametrize("f", [2, 16, 128, 260])
@pytest.mark.parametrize("nslices", [1, 8, 3])
def test_spmm_split_long_rows_deterministic(K, dev, f, nslices):
    """Rows longer than the plan chunk go through partial slots + the fix-up kernel."""
    synth = pkg("synth")
    n, row, col, val = synth.make_graph(4000, 400000, seed=2)
    A = sp.csr_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    assert np.diff(A.indptr).max() > 600
    rng = np.random.default_rng(0)
    B = rng.random((n, f), dtype=np.float32) * 2 - 1
    ref = oracle.spmm(A, B)
    got, d = _spmm_gpu(K, dev, A, B, chunk=128, nslices=nslices)
    assert d.nfix > 0 and d.nslots > d.nfix
    assert rel_err(got, ref) < TOL
    got2, _ = _spmm_gpu(K, dev, A, B, chunk=128, nslices=nslices)
    np.testing.assert_array_equal(got, got2)                      # no atomics => bit-reproducible
    base = rng.random((n, f), dtype=np.float32)
    acc, _ = _spmm_gpu(K, dev, A, B, chunk=128, accumulate_into=base, nslices=nslices)
    assert rel_err(acc, ref + base) < TOL
    acc1, _ = _spmm_gpu(K, dev, A, B, accumulate_into=base, nslices=nslices)   # unsplit accumulate
    assert rel_err(acc1, ref + base) < TOL


@pytest.mark.parametrize("f", [4, 16, 30, 64, 128, 132, 256])
@pytest.mark.parametrize("nslices", [1, 8])
def test_spmm_dense_core_lds_kernel(K, dev, f, nslices):
    """Degree-sorted graph: dense 128x128 tiles go through the LDS-tiled kernel, the rest through
    the gather kernel, one combined fix-up.  Same tolerance as every other SpMM path."""
    partition, synth = pkg("partition"), pkg("synth")
    n, row, col, val = synth.make_graph(4000, 400000, seed=2)
    deg = torch.bincount(row, minlength=n)
    rank = torch.empty(n, dtype=torch.int64)
    rank[torch.argsort(-deg, stable=True)] = torch.arange(n)
    r, c = rank[row], rank[col]
    A = sp.csr_matrix((val.numpy(), (r.numpy(), c.numpy())), shape=(n, n))
    h = partition.csr_from_coo(r, c, val, n, n, nslices=nslices, core=True, tau=0.05, emax=6000,
                               ngroups=3 if nslices > 1 else None, strip=False)
    assert h.core is not None and h.core.nnz > 0.2 * A.nnz and h.ngroups == (3 if nslices > 1 else 1)
    d = K.prepare(h)
    assert d.core is not None
    assert d.nslots_total == d.nslots + h.core.nslots + (h.dense3.nslots if h.dense3 is not None else 0)
    rng = np.random.default_rng(f)
    B = rng.random((n, f), dtype=np.float32) * 2 - 1
    ref = oracle.spmm(A, B)
    Bd = torch.from_numpy(B).to(dev)
    C = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, Bd, C)
    torch.cuda.synchronize()
    got = C.cpu().numpy()
    assert rel_err(got, ref) < TOL
    C2 = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, Bd, C2)
    assert torch.equal(C, C2)                                    # deterministic
    base = rng.random((n, f), dtype=np.float32)
    C3 = torch.from_numpy(base).to(dev)
    K.spmm(d, Bd, C3, accumulate=True)
    assert rel_err(C3.cpu().numpy(), ref + base) < TOL
    # Inf in a feature row that no entry references must not leak through the staged panels
    free = np.setdiff1d(np.arange(n), A.indices)
    if free.size:
        B2 = B.copy(); B2[free[0]] = np.inf
        C4 = torch.empty((n, f), device=dev)
        K.spmm(d, torch.from_numpy(B2).to(dev), C4)
        assert np.isfinite(C4.cpu().numpy()).all()


@pytest.mark.parametrize("f", [4, 30, 64, 100, 128, 132, 256])
@pytest.mark.parametrize("nslices", [1, 8])
def test_spmm_bf16x3_blocks(K, dev, f, nslices):
    """The densest 512 x 128 blocks go through the bf16 matrix cores with the three-plane split
    (pgcn_spmm_dense_bf16x3_f32), the rest through strips / the gather kernel; one fix-up.  Same tolerance as every
    other SpMM path -- and the per-row bound against float64 (the split is fp32-accurate, not bf16-accurate), bit-equal
    repeats, accumulate, an unaligned operand, structural zeros under Inf, a partial last panel and block row."""
    partition = pkg("partition")
    rng = np.random.default_rng(300 + f + nslices)
    n, m = 1300, 700                                         # 3 block rows (the last one 276 rows), 6 panels (the last one 60 columns)
    D = (rng.random((n, m)) < 0.004).astype(np.float32)
    D[:512, :384] = rng.random((512, 384)) < 0.45            # three full blocks in block row 0
    D[512:1024, :128] = rng.random((512, 128)) < 0.25        # one in block row 1
    D[1024:, 640:] = rng.random((276, 60)) < 0.9             # the partial corner block: 276 x 60 of 512 x 128 = 23 % full
    D[5, :] = 0                                              # an empty row inside a block
    D[:, 300] = 0                                            # a column nobody references
    D *= (rng.standard_normal((n, m)) * np.exp(rng.standard_normal((n, 1)))).astype(np.float32)      # rows of different scale
    A = sp.csr_matrix(D)
    h = partition.csr_from_scipy(A, nslices=nslices, core=True, strip=True, strip_min=32, dense3_tau=0.2)
    assert h.dense3 is not None and h.dense3.blk_row.tolist() == [0, 0, 0, 1, 2] and h.nnz == A.nnz
    d = K.prepare(h)
    assert d.dense3 is not None and d.nslots_total >= d.nslots + h.dense3.nslots
    B = (rng.random((m, f), dtype=np.float32) * 2 - 1) * np.exp(rng.standard_normal((m, 1))).astype(np.float32)
    ref = oracle.spmm(A, B)
    Bd = torch.from_numpy(B).to(dev)
    C = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, Bd, C)
    torch.cuda.synchronize()
    got = C.cpu().numpy()
    assert rel_err(got, ref) < TOL
    ref64 = A.astype(np.float64) @ B.astype(np.float64)
    bound = abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    worst = float((np.abs(got - ref64) / (1e-5 * bound + 1e-30)).max())
    assert worst <= 1.0, "a row exceeds 1e-5 * sum|a||x| by a factor %.3g" % worst
    C2 = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, Bd, C2)
    assert torch.equal(C, C2)                                # deterministic
    base = rng.random((n, f), dtype=np.float32)
    C3 = torch.from_numpy(base).to(dev)
    K.spmm(d, Bd, C3, accumulate=True)
    assert rel_err(C3.cpu().numpy(), ref + base) < TOL
    wide = torch.zeros((m, f + 3), device=dev)               # an odd leading dimension / unaligned base
    wide[:, 1:f + 1] = Bd
    C4 = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, wide[:, 1:f + 1], C4)
    assert rel_err(C4.cpu().numpy(), ref) < TOL
    # Inf in a feature row no entry references: nothing leaks.  Inf in a referenced row: exactly the rows with an
    # entry in that column see it (the exact path multiplies only where A != 0).
    B2 = B.copy(); B2[300] = np.inf
    C5 = torch.empty((n, f), device=dev)
    K.spmm(d, torch.from_numpy(B2).to(dev), C5)
    assert np.isfinite(C5.cpu().numpy()).all() and rel_err(C5.cpu().numpy(), ref) < TOL
    B3 = B.copy(); B3[17, 0] = np.inf
    C6 = torch.empty((n, f), device=dev)
    K.spmm(d, torch.from_numpy(B3).to(dev), C6)
    got = C6.cpu().numpy()
    hit = np.asarray(A[:, 17].todense()).ravel() != 0
    assert hit.sum() > 100                                   # column 17 crosses the blocks
    assert np.isinf(got[hit, 0]).all() and np.isfinite(got[~hit]).all() and np.isfinite(got[:, 1:]).all()
    assert rel_err(got[:, 1:], ref[:, 1:]) < TOL


@pytest.mark.parametrize("lanes", ["strip/gather+dense3", "strip/dense3/gather", "dense3+strip/gather", "gather+strip+dense3"])
def test_spmm_launch_lanes_bit_identical(K, dev, lanes, monkeypatch):
    """tuning.lanes: the producers of one product on two or three streams write the same partial rows and the fix-up adds
    them in the same order -- the result equals the one-stream result bit for bit, launched eagerly, back to back on the
    same work-space, and as a replayed HIP graph (fork and join are events inside the capture)."""
    partition, tuning = pkg("partition"), pkg("tuning")
    rng = np.random.default_rng(77)
    n, m, f = 2100, 900, 128
    D = (rng.random((n, m)) < 0.02).astype(np.float32)
    D[:1024, :256] = rng.random((1024, 256)) < 0.5           # bf16 blocks
    D[:1536, 256:640] = rng.random((1536, 384)) < 0.08       # strip tiles
    D *= rng.standard_normal((n, m)).astype(np.float32)
    A = sp.csr_matrix(D)
    h = partition.csr_from_scipy(A, nslices=8, core=True, strip=True, strip_min=32, dense3_tau=0.2)
    assert h.dense3 is not None and h.strip is not None and h.nnz == A.nnz
    d = K.prepare(h)
    assert d.ntasks > 0
    Bd = torch.from_numpy(rng.standard_normal((m, f)).astype(np.float32)).to(dev)
    B2 = torch.from_numpy(rng.standard_normal((m, f)).astype(np.float32)).to(dev)
    K.single_lane = True
    try:
        want, want2 = torch.empty((n, f), device=dev), torch.empty((n, f), device=dev)
        K.spmm(d, Bd, want); K.spmm(d, B2, want2)
    finally:
        K.single_lane = False
    assert rel_err(want.cpu().numpy(), oracle.spmm(A, Bd.cpu().numpy())) < TOL
    monkeypatch.setattr(tuning.T, "lanes", lanes)
    monkeypatch.setattr(tuning.T, "lanes_min_nnz", 0)
    d.launch_cache.clear()
    nside = lanes.count("/")
    got, got2 = torch.full((n, f), float("nan"), device=dev), torch.full((n, f), float("nan"), device=dev)
    for _ in range(3):                                       # back to back: the second product re-uses the first one's work-space
        K.spmm(d, Bd, got); K.spmm(d, B2, got2)
    torch.cuda.synchronize()
    assert len(K.sides) >= nside
    assert torch.equal(got, want) and torch.equal(got2, want2)
    g = torch.cuda.CUDAGraph()
    cg, cg2 = torch.zeros((n, f), device=dev), torch.zeros((n, f), device=dev)
    with torch.cuda.graph(g):
        K.spmm(d, Bd, cg); K.spmm(d, B2, cg2)
    for _ in range(2):
        cg.fill_(float("nan")); cg2.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(cg, want) and torch.equal(cg2, want2)
    d.launch_cache.clear()


@pytest.mark.parametrize("f", [4, 30, 64, 128, 132, 256])
@pytest.mark.parametrize("nslices", [1, 8])
def test_spmm_strip_tiles(K, dev, f, nslices):
    """512 x 128 strip tiles through the asynchronously staged LDS pipeline (pgcn_spmm_strip_f32): tiles of
    one record, tiles cut into several records, several pieces per tile row, the densest 128 x 128 tiles on
    the matrix cores and a sparse rest in the gather kernel; one fix-up.  Same tolerance as every SpMM path."""
    partition = pkg("partition")
    rng = np.random.default_rng(100 + f + nslices)
    n, m = 1300, 700
    D = (rng.random((n, m)) < 0.0007).astype(np.float32)            # sparse rest: ~45 entries per strip tile
    D[:512, :128] = rng.random((512, 128)) < 0.08                    # 5 k entries -> 4 records of one tile
    D[:512, 128:256] = rng.random((512, 128)) < 0.01                 # one record
    D[512:1024, 256:640] = rng.random((512, 384)) < 0.03             # three panels of the second tile row
    D[1024:1152, :128] = rng.random((128, 128)) < 0.5                # MFMA tile inside the third (ragged) tile row
    D[1152:, 384:512] = rng.random((148, 128)) < 0.2                 # ragged last tile row (rows 1024..1299)
    D[:512, 640:] = rng.random((512, 60)) < 0.05                     # last, partial column block: the windowed panel [572, 700)
    D[7, :] = 0                                                      # an empty row inside strip tiles
    D[:, 300] = 0                                                    # a column nobody references
    D *= rng.standard_normal((n, m)).astype(np.float32)
    A = sp.csr_matrix(D)
    h = partition.csr_from_scipy(A, nslices=nslices, core=True, dense3_tau=2.0, strip=True, strip_min=64)
    assert h.strip is not None and h.core is None and h.dense3 is None
    assert int(h.strip.rec[:, 3].max()) >= 4 and h.strip.rec.shape[0] >= 9      # tiles of several layers
    assert h.strip.nnz > 0.4 * A.nnz and h.col.numel() > 0
    assert h.nnz == A.nnz
    d = K.prepare(h)
    assert d.strip is not None and d.nslots_total == d.nslots + h.strip.nslots
    B = rng.random((m, f), dtype=np.float32) * 2 - 1
    ref = oracle.spmm(A, B)
    Bd = torch.from_numpy(B).to(dev)
    C = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, Bd, C)
    torch.cuda.synchronize()
    assert rel_err(C.cpu().numpy(), ref) < TOL
    C2 = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, Bd, C2)
    assert torch.equal(C, C2)                                        # deterministic
    base = rng.random((n, f), dtype=np.float32)
    C3 = torch.from_numpy(base).to(dev)
    K.spmm(d, Bd, C3, accumulate=True)
    assert rel_err(C3.cpu().numpy(), ref + base) < TOL
    # odd leading dimension / unaligned base: the plain (unstaged) strip kernel
    wide = torch.zeros((m, f + 3), device=dev)
    wide[:, 1:f + 1] = Bd
    C4 = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, wide[:, 1:f + 1], C4)
    assert rel_err(C4.cpu().numpy(), ref) < TOL
    # Inf in a row of B no entry references must not leak out of a staged panel; Inf in a referenced row
    # reaches exactly the rows that have an entry in that column
    B2 = B.copy(); B2[300] = np.inf
    C5 = torch.empty((n, f), device=dev)
    K.spmm(d, torch.from_numpy(B2).to(dev), C5)
    assert np.isfinite(C5.cpu().numpy()).all() and rel_err(C5.cpu().numpy(), ref) < TOL
    B3 = B.copy(); B3[17, 0] = np.inf
    C6 = torch.empty((n, f), device=dev)
    K.spmm(d, torch.from_numpy(B3).to(dev), C6)
    got = C6.cpu().numpy()
    hit = np.asarray(A[:, 17].todense()).ravel() != 0
    assert hit.sum() > 40
    assert np.isinf(got[hit, 0]).all() and np.isfinite(got[~hit]).all() and np.isfinite(got[:, 1:]).all()
    assert rel_err(got[:, 1:], ref[:, 1:]) < TOL


def test_spmm_feature_passes_bit_identical(K, dev):
    """The one opt-in shape of the launch group leaves the SAME bits as the default one: the gather kernel in 64-
    feature passes (PGCN_SPMM_FPASS64, automatic on whole graphs) next to strips, MFMA tiles and sliced hub rows."""
    partition, _lib = pkg("partition"), pkg("_lib")
    rng = np.random.default_rng(77)
    n, m, f = 2100, 1500, 128
    D = (rng.random((n, m)) < 0.004).astype(np.float32)
    D[:1024, :256] = rng.random((1024, 256)) < 0.05                  # strip tiles of several layers, two tile rows
    D[1024:1536, 1408:] = rng.random((512, 92)) < 0.06               # ... one of them in the last, partial column block
    D[:128, 256:384] = rng.random((128, 128)) < 0.6                  # an MFMA tile
    D[40, :] = rng.random(m) < 0.7                                   # a hub row: several gather tasks per slice
    D *= rng.standard_normal((n, m)).astype(np.float32)
    A = sp.csr_matrix(D)
    h = partition.csr_from_scipy(A, nslices=8, core=True, dense3_tau=2.0, strip=True, strip_min=64)
    assert h.strip is not None and h.col.numel() > 0
    d = K.prepare(h)
    Bd = torch.from_numpy(rng.random((m, f), dtype=np.float32) * 2 - 1).to(dev)
    flags0 = K.base_flags
    try:
        K.base_flags = flags0 & ~_lib.SPMM_FPASS64
        d.launch_cache.clear()
        ref = torch.full((n, f), float("nan"), device=dev)
        K.spmm(d, Bd, ref)
        assert rel_err(ref.cpu().numpy(), oracle.spmm(A, Bd.cpu().numpy())) < TOL
        K.base_flags = flags0 | _lib.SPMM_FPASS64
        d.launch_cache.clear()
        C = torch.full((n, f), float("nan"), device=dev)
        K.spmm(d, Bd, C)
        torch.cuda.synchronize()
        assert torch.equal(C, ref)
    finally:
        K.base_flags = flags0
        d.launch_cache.clear()


def test_spmm_strip_pieces_and_panel_reuse(K, dev):
    """A tile row whose records are cut into several pieces, with consecutive records (layers of one tile)
    sharing a staged panel inside a piece (flag bit 0) and piece boundaries falling between layers of a tile."""
    partition = pkg("partition")
    rng = np.random.default_rng(5)
    n, m, f = 512, 1024, 128
    D = (rng.random((n, m)) < 0.06).astype(np.float32) * rng.standard_normal((n, m)).astype(np.float32)
    A = sp.csr_matrix(D)
    r, c = torch.from_numpy(A.tocoo().row.astype(np.int64)), torch.from_numpy(A.tocoo().col.astype(np.int64))
    v = torch.from_numpy(A.tocoo().data.astype(np.float32))
    keep, st = partition.build_strips(r, c, v, n, m, min_entries=1, pieces=5, layer_min=1)
    assert st is not None and not bool(keep.any())
    assert 3 <= st.npieces <= 8 and int(st.rec[:, 1].sum()) > 0       # some records reuse the staged panel
    first_recs = st.work[:, 1].long()
    assert bool((st.rec[first_recs, 1] == 0).all())                   # ... but never the first of a piece
    h = partition.csr_from_scipy(A, nslices=1, core=True, dense3_tau=2.0, strip=True, strip_min=1)
    d = K.prepare(h)
    B = rng.random((m, f), dtype=np.float32) * 2 - 1
    C = torch.full((n, f), float("nan"), device=dev)
    K.spmm(d, torch.from_numpy(B).to(dev), C)
    assert rel_err(C.cpu().numpy(), oracle.spmm(A, B)) < TOL


def test_spmm_edge_cases(K, dev):
    # empty rows, empty matrix, single row, explicit zeros, duplicate-free pattern (val=None)
    A = sp.csr_matrix((np.array([1., 2., 0., 3.], np.float32), np.array([0, 3, 1, 2], np.int32),
                       np.array([0, 2, 2, 2, 4], np.int64)), shape=(4, 4))
    B = np.arange(16, dtype=np.float32).reshape(4, 4)
    got, _ = _spmm_gpu(K, dev, A, B)
    np.testing.assert_array_equal(got, A.toarray() @ B)           # exact in fp32 (small ints)
    Z = sp.csr_matrix((3, 4), dtype=np.float32)
    got, _ = _spmm_gpu(K, dev, Z, B)
    np.testing.assert_array_equal(got, np.zeros((3, 4), np.float32))
    h = _host_csr(A)
    d = K.prepare(h, pattern_only=True)
    C = torch.empty((4, 4), device=dev)
    K.spmm(d, torch.from_numpy(B).to(dev), C)
    np.testing.assert_array_equal(C.cpu().numpy(), (A.toarray() != 0).astype(np.float32) @ B +
                                  np.array([[0] * 4, [0] * 4, [0] * 4, B[1]]))  # stored zero counts as 1
    # NaN / Inf in rows that are NOT referenced must not leak (no 0*Inf from padding lanes)
    B2 = B.copy(); B2[0, :] = np.inf
    A2 = sp.csr_matrix((np.array([1.], np.float32), np.array([2], np.int32), np.array([0, 1, 1], np.int64)),
                       shape=(2, 4))
    got, _ = _spmm_gpu(K, dev, A2, B2)
    np.testing.assert_array_equal(got, np.vstack([B2[2], np.zeros(4, np.float32)]))


def test_spmm_row_map(K, dev):
    partition = pkg("partition")
    A = sp.csr_matrix(mmread(gpath("gemat11p.A.mtx"))).astype(np.float32)
    n = A.shape[0]
    rows = np.sort(np.random.default_rng(0).permutation(n)[:1500]).astype(np.int32)
    sub = _host_csr(A[rows])
    sub.row_map = torch.from_numpy(rows)
    rng = np.random.default_rng(2)
    B = rng.random((n, 32), dtype=np.float32)
    base = rng.random((n, 32), dtype=np.float32)
    for chunk in (8, 4096):
        old, K.chunk = K.chunk, chunk
        d = K.prepare(sub)
        K.chunk = old
        C = torch.from_numpy(base).to(dev)
        K.spmm(d, torch.from_numpy(B).to(dev), C, accumulate=True)
        ref = base.copy()
        ref[rows] += oracle.spmm(A[rows], B)
        assert rel_err(C.cpu().numpy(), ref) < TOL
        untouched = np.setdiff1d(np.arange(n), rows)
        np.testing.assert_array_equal(C.cpu().numpy()[untouched], base[untouched])


@pytest.mark.parametrize("f", [1, 4, 6, 128, 516])
def test_gather_scatter_bit_exact(K, dev, f):
    rng = np.random.default_rng(f)
    H = rng.random((5000, f), dtype=np.float32)
    idx = rng.permutation(5000)[:1733].astype(np.int32)
    Hd, idxd = torch.from_numpy(H).to(dev), torch.from_numpy(idx).to(dev)
    out = torch.empty((idx.size, f), device=dev)
    K.gather_rows(Hd, idxd, out)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.gather_rows(H, idx))
    src = rng.random((idx.size, f), dtype=np.float32)
    for acc in (False, True):
        X = torch.from_numpy(H).to(dev)
        K.scatter_rows(X, idxd, torch.from_numpy(src).to(dev), acc)
        ref = H.copy()
        oracle.scatter_rows(ref, idx, src, acc)
        np.testing.assert_array_equal(X.cpu().numpy(), ref)
    K.gather_rows(Hd, idxd[:0], out)                               # empty index list is a no-op
    # accumulate with REPEATED indices (a boundary row that comes back from several peers): every row is added
    dup = np.concatenate([idx[:400], idx[:400], idx[:100]]).astype(np.int32)
    srcd = rng.random((dup.size, f), dtype=np.float32)
    X = torch.from_numpy(H).to(dev)
    K.scatter_rows(X, torch.from_numpy(dup).to(dev), torch.from_numpy(srcd).to(dev), True)
    ref = H.astype(np.float64)
    np.add.at(ref, dup, srcd.astype(np.float64))
    assert rel_err(X.cpu().numpy(), ref) < 1e-6


class _PrecomputedExchanger:
    """Test double for one GPU: RCCL refuses two ranks on one device, so the slab a rank
    would receive is produced from the global matrix the test already holds."""

    def __init__(self):
        self.next_recv = None
        self.sent_parts, self.cursor = [], 0

    def begin(self, recv_rows):
        """Arm one exchange: `recv_rows` is the whole slab the rank should receive (all rounds)."""
        self.next_recv, self.sent_parts, self.cursor = recv_rows, [], 0

    @property
    def sent(self):
        return torch.cat(self.sent_parts) if self.sent_parts else None

    def alltoallv(self, send, send_off, recv, recv_off, f):   # called once per round, sub-slabs
        self.sent_parts.append(send[:send_off[-1]].clone())
        k = recv_off[-1]
        recv[:k] = self.next_recv[self.cursor:self.cursor + k]
        self.cursor += k

    def allreduce_sum(self, buf):
        pass


def _virtual_ranks_fwd_bwd(K, dev, A, part, P, Hfull, Gfull):
    """Every rank's engine on the one GPU (the exchange is emulated from the global matrices the
    test holds).  Returns (forward A.H, backward A^T.G, engines) in GLOBAL row numbering."""
    partition, engine = pkg("partition"), pkg("engine")
    n, f = Hfull.shape
    row, col, val = (torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64)),
                     torch.from_numpy(A.data.astype(np.float32)))
    fwd = np.zeros((n, f), np.float32)
    engines = []
    for r in range(P):
        p = partition.build_partition(row, col, val, n, part, r, P)
        ex = _PrecomputedExchanger() if P > 1 else None
        eng = engine.AggregationEngine(p, K, dev, ex)
        if P > 1:
            ex.begin(torch.from_numpy(Hfull[p.halo_global.numpy()]).to(dev))
        own = p.owned.numpy()
        out = eng.forward(torch.from_numpy(Hfull[own]).to(dev))
        torch.cuda.synchronize()
        fwd[own] = out.cpu().numpy()
        if P > 1:   # what was packed for the peers is exactly H[send rows], in the peer's slab order
            np.testing.assert_array_equal(ex.sent.cpu().numpy(), Hfull[p.send_global.numpy()])
        engines.append((eng, ex, p))
    # backward: first pass collects every rank's halo partials, second pass delivers them
    partials = {}
    for eng, ex, p in engines:
        if P > 1:
            ex.begin(torch.zeros((p.n_send, f), device=dev))
        eng.backward(torch.from_numpy(Gfull[p.owned.numpy()]).to(dev))
        torch.cuda.synchronize()
        if P > 1:   # what this rank computed for rows owned by others: (owner, global id) -> partial row
            partials[p.rank] = (p.halo_owner.numpy(), p.halo_global.numpy(), ex.sent.cpu().numpy())
    bwd = np.zeros((n, f), np.float32)
    for eng, ex, p in engines:
        if P > 1:
            back = np.zeros((p.n_send, f), np.float32)
            tgt, gid = p.send_owner.numpy(), p.send_global.numpy()
            for q in range(P):
                if q == p.rank:
                    continue
                ho, hg, slab = partials[q]
                mine = ho == p.rank                                # what q computed for my rows, q's slab order
                pos = np.nonzero(tgt == q)[0]                      # where q's rows sit in my send slab
                np.testing.assert_array_equal(hg[mine], gid[pos])  # same (round, degree-rank) order on both sides
                back[pos] = slab[mine]
            ex.begin(torch.from_numpy(back).to(dev))
        out = eng.backward(torch.from_numpy(Gfull[p.owned.numpy()]).to(dev))
        torch.cuda.synchronize()
        bwd[p.owned.numpy()] = out.cpu().numpy()
    return fwd, bwd, engines


@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES + SPMM_CASES_MORE)
def test_engine_forward_backward_vs_reference_golden(K, dev, name, mtx, pv, P):
    """Every rank's engine on the one GPU (real kernels), against the outputs of the reference's own PSpMM
    (GPU/PGCN.py:121-134) -- the Cora shape of BASELINE configs[0] with two and four ranks included.  Both sides are
    measured against a float64 product of the same matrices (conftest.held_to_fixture): no blanket tolerance."""
    arrays, meta = golden(name)
    A = sp.coo_matrix(mmread(gpath(mtx)))
    n, f = A.shape[0], meta["f"]
    part = torch.tensor(read_partvec(gpath(pv)))
    Hfull, Gfull = golden_inputs(n, f, meta["seed"])
    fwd, bwd, _ = _virtual_ranks_fwd_bwd(K, dev, A, part, P, Hfull, Gfull)
    A64 = sp.csr_matrix(A).astype(np.float64)                     # (duplicates add, like the reference's uncoalesced COO)
    ref_f, ref_b = A64 @ Hfull.astype(np.float64), A64.T.tocsr() @ Gfull.astype(np.float64)
    held_to_fixture(name, "PSpMM.forward", fwd, arrays["fwd"], ref_f)
    assert rel_err(fwd, arrays["fwd"]) < 2 * TOL                  # ... and to each other, both being within the floor
    assert rel_err(bwd, oracle.spmm(sp.csr_matrix(A.T).astype(np.float32), Gfull)) < TOL
    if "bwd" in arrays:
        held_to_fixture(name, "PSpMM.backward", bwd, arrays["bwd"], ref_b)
    else:
        assert rel_err(bwd, ref_b) < TOL


@pytest.mark.parametrize("P,f", [(2, 128), (3, 64)])
def test_engine_halo_dense_core(K, dev, P, f, monkeypatch):
    """Multi-rank partition of a power-law graph: the degree-ordered receive slab gives A_halo and
    A_halo^T dense tiles too, so the LDS-tiled kernel runs on the halo pass as well."""
    synth, partition = pkg("synth"), pkg("partition")
    monkeypatch.setattr(partition, "CORE_MIN_NNZ", 0)            # keep the (small) cores of this small graph
    monkeypatch.setattr(partition, "CORE_MIN_FRAC", 0.0)
    n, row, col, val = synth.make_graph(6000, 900000, seed=3)
    A = sp.coo_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    part = synth.random_partvec(n, P, seed=1)
    rng = np.random.default_rng(5)
    Hfull = rng.random((n, f), dtype=np.float32) * 2 - 1
    Gfull = rng.random((n, f), dtype=np.float32) * 2 - 1
    fwd, bwd, engines = _virtual_ranks_fwd_bwd(K, dev, A, part, P, Hfull, Gfull)
    assert any((a.core or a.strip) is not None for e, _, _ in engines for a in e.A_halo)
    assert any((a.core or a.strip) is not None for e, _, _ in engines for a in e.A_halo_T)
    assert all(e.rounds == 2 for e, _, _ in engines)
    Ac = sp.csr_matrix(A)
    assert rel_err(fwd, oracle.spmm(Ac, Hfull)) < TOL
    assert rel_err(bwd, oracle.spmm(sp.csr_matrix(A.T), Gfull)) < TOL
    # power-law rows span three orders of magnitude: the PER-ROW bound of tests/test_fullsize_gpu.py against a
    # float64 shadow, |got_i - ref_i| <= 1e-5 * sum_j |a_ij| |x_j| element-wise (a wrong low-degree row cannot hide
    # behind a hub row's magnitude)
    A64, absA = Ac.astype(np.float64), abs(Ac).astype(np.float64)
    for got, M, Ma, X in ((fwd, A64, absA, Hfull), (bwd, A64.T.tocsr(), absA.T.tocsr(), Gfull)):
        ref64, bound = M @ X.astype(np.float64), Ma @ np.abs(X).astype(np.float64)
        worst = float((np.abs(got.astype(np.float64) - ref64) / (1e-5 * bound + 1e-30)).max())
        assert worst <= 1.0, "a row exceeds 1e-5 * sum|a||x| by a factor %.3g" % worst


@pytest.mark.parametrize("name,mtx,pv", TRAIN_CASES + TRAIN_CASES_MORE)
def test_run_matches_reference_training(dev, name, mtx, pv):
    """The drop-in's run() on the GPU at P=1 vs losses/weights of the reference's run() (GPU/PGCN.py:194-226; the Cora
    shape of BASELINE configs[0] included).  Five Adam steps divide by sqrt(v): where a gradient entry is near zero the
    reference's own fp32 run is already far from exact arithmetic, so both runs are measured against the float64
    shadow of the loop (oracle.pgcn_train_np) and the HIP run may be as far as the floor 1e-5 or twice the
    reference's own distance (conftest.held_to_fixture)."""
    arrays, meta = golden(name)
    M = pkg("PGCN")
    M._kernel_provider = None
    M._exchanger = None
    torch.manual_seed(meta["seed"])
    w0 = [torch.nn.Linear(meta["f"], meta["f"], bias=False).weight.detach().numpy() for _ in range(meta["nlayers"])]
    for i, w in enumerate(w0):                                     # same RNG stream as the golden run
        np.testing.assert_array_equal(w, arrays["w0_%d" % i])
    torch.manual_seed(meta["seed"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model = M.run(0, 1, meta["nlayers"], meta["f"], gpath(mtx), gpath(pv), "nccl")
    printed = [float(x) for x in re.findall(r"Epoch \d{5} \| Loss ([0-9.]+)", buf.getvalue())]
    assert type(M._kernel_provider).__name__ == "HipKernels"
    A = sp.csr_matrix(mmread(gpath(mtx)))
    n, f = A.shape[0], meta["f"]
    H0 = np.repeat(np.arange(n, dtype=np.float64)[:, None], f, axis=1)            # PGCN.py:187-189
    losses64, W64 = oracle.pgcn_train_np(A, [0] * n, 1, w0, H0, np.arange(n) % f, epochs=5)
    # the printed loss has four decimals ("{:.4f}", PGCN.py:224): half a unit of the last digit is the print's own error
    e_fix = rel_err(arrays["losses"][1:], losses64[1:])
    bound = max(1e-5, 2 * e_fix) * np.abs(losses64[1:]) + 0.5e-4
    assert (np.abs(np.array(printed) - losses64[1:]) <= bound).all(), (printed, losses64[1:], e_fix)
    for i, m in enumerate(model):
        held_to_fixture(name, "run() weight %d after 5 Adam steps" % i, m.linear.weight.detach().cpu().numpy(),
                        arrays["w1_%d" % i], W64[i])


def test_pargcn_semantics_vs_oracle(K, dev):
    """Parallel-GCN/main.c training loop with both aggregations on the HIP engine."""
    partition, engine, pargcn = pkg("partition"), pkg("engine"), pkg("pargcn")
    A = oracle.normalize_adjacency(mmread(gpath("gemat11p.mtx")))
    A = ((A + A.T) * 0.5).tocoo().astype(np.float32)
    n = A.shape[0]
    d = [n, 16, 16, 2]
    rng = np.random.default_rng(3)
    W = {l: (rng.random((d[l], d[l + 1]), dtype=np.float32) * 2 - 1) * np.float32(np.sqrt(6.0 / (d[l] + d[l + 1])))
         for l in (1, 2)}
    Y = np.zeros((n, 2), np.float32); Y[:, 1] = 1
    Ym = np.zeros((n, 2), np.uint8); Ym[:, 1] = 1
    err, Wc, Hl, _ = oracle.pargcn_train(A, [0] * n, 1, d, W, np.ones((n, 16), np.float32), Y, Ym)
    p = partition.build_partition(torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64)),
                                  torch.from_numpy(A.data), n, torch.zeros(n, dtype=torch.int64), 0, 1)
    eng = engine.AggregationEngine(p, K, dev)
    errs, Wn, Hout = pargcn.train(eng, d, {l: torch.from_numpy(w).to(dev) for l, w in W.items()},
                                  torch.ones((n, 16), device=dev), torch.from_numpy(Y).to(dev),
                                  torch.from_numpy(Ym).to(dev))
    np.testing.assert_allclose(errs, err, rtol=1e-5)
    for l in (1, 2):
        assert rel_err(Wn[l].cpu().numpy(), Wc[l]) < TOL
    assert rel_err(Hout.cpu().numpy(), Hl[p.owned.numpy()]) < TOL     # local row i = global row owned[i]


def test_full_size_properties_reddit_like(K, dev):
    """BASELINE size (n = 232 965, ~114.8 M stored entries, f = 128): size-independent
    properties + the oracle on a row sample."""
    synth, partition, engine = pkg("synth"), pkg("partition"), pkg("engine")
    n, row, col, val = synth.make_graph("reddit", seed=0, device=dev)
    nnz = row.numel()
    assert nnz == 114615892 + n
    p = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64), 0, 1)
    eng = engine.AggregationEngine(p, K, dev)
    f = 128
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    X = torch.rand(n, f, device=dev, generator=gen) * 2 - 1
    Yv = torch.rand(n, f, device=dev, generator=gen) * 2 - 1
    AX, AY = eng.forward(X), eng.forward(Yv)
    # determinism
    assert torch.equal(AX, eng.forward(X))
    # linearity: A(2X - 3Y) = 2AX - 3AY
    lin = eng.forward(2 * X - 3 * Yv)
    assert float((lin - (2 * AX - 3 * AY)).abs().max() / lin.abs().max()) < 2e-5
    # checksum of checksums: 1^T (A X) = (A^T 1)^T X, column sums in float64
    ones = torch.ones(n, 4, device=dev)
    At1 = eng.backward(ones)[:, 0].double()
    lhs = AX.double().sum(0)
    rhs = (At1.unsqueeze(1) * X.double()).sum(0)
    assert float((lhs - rhs).abs().max() / lhs.abs().max()) < 1e-5
    # row sums of A_hat: A.1 computed by the kernel vs a float64 segment sum of the values
    rs = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, row, val.double())
    rs = rs[p.owned.to(dev)]                      # local row i = global row owned[i]
    assert float((eng.forward(ones)[:, 0].double() - rs).abs().max() / rs.max()) < 1e-5
    # symmetric matrix => backward operand gives the same product
    assert float((eng.backward(X) - AX).abs().max() / AX.abs().max()) < 2e-5
    # the oracle on a sample of rows (full CSR copied to the host once)
    rows = np.sort(np.random.default_rng(0).choice(n, 600, replace=False)).astype(np.int32)
    rp, ci, va = (t.cpu().numpy() for t in partition.full_csr(p.A_loc))
    Xh = X.cpu().numpy()
    ref = np.zeros((n, f), np.float32)
    L = oracle.lib()
    L.oracle_spmm_csr_rows_f32(rows.size, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                               rp.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                               ci.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                               va.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                               Xh.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), f,
                               ref.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), f, f, 0)
    assert rel_err(AX.cpu().numpy()[rows], ref[rows]) < TOL


def test_rccl_single_rank_comm(dev):
    """C-ABI RCCL entry points on a 1-rank communicator (RCCL allows one rank per device)."""
    _lib = pkg("_lib")
    L = _lib.lib()
    uid = ctypes.create_string_buffer(128)
    _lib.check(L.pgcn_comm_unique_id(uid), "unique_id")
    comm = ctypes.c_void_p()
    _lib.check(L.pgcn_comm_init(ctypes.byref(comm), uid, 1, 0), "comm_init")
    buf = torch.arange(1000, dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.pgcn_allreduce_sum_f32(comm, buf.data_ptr(), buf.numel(), s), "allreduce")
    off = (ctypes.c_int64 * 2)(0, 0)
    _lib.check(L.pgcn_exchange_alltoallv_f32(comm, None, off, None, off, 128, s), "alltoallv")
    bad = (ctypes.c_int64 * 2)(0, 3)
    assert L.pgcn_exchange_alltoallv_f32(comm, None, bad, None, off, 128, s) == -1   # own segment non-empty
    torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.float32))
    _lib.check(L.pgcn_comm_destroy(comm), "comm_destroy")


def test_rccl_calls_are_capturable_in_a_hip_graph(dev):
    """bench.py --graph captures a whole N-rank training step, the RCCL calls of the boundary exchange and of
    average_gradients included.  What one GPU can show of that: the library's all-reduce and (peer-less) all-to-all-v
    on a ONE-rank communicator record into a capturing stream without error and replay (Parallel-GCN/main.c:321,425
    MPI_Allreduce; GPU/PGCN.py:99-115 send / recv)."""
    _lib = pkg("_lib")
    L = _lib.lib()
    uid = ctypes.create_string_buffer(128)
    _lib.check(L.pgcn_comm_unique_id(uid), "unique_id")
    comm = ctypes.c_void_p()
    _lib.check(L.pgcn_comm_init(ctypes.byref(comm), uid, 1, 0), "comm_init")
    buf = torch.arange(4096, dtype=torch.float32, device=dev)
    out = torch.zeros_like(buf)
    off = (ctypes.c_int64 * 2)(0, 0)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):                       # one eager call first: lazy initialisation must not happen inside a capture
        _lib.check(L.pgcn_allreduce_sum_f32(comm, buf.data_ptr(), buf.numel(), side.cuda_stream), "allreduce (eager)")
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream(dev).cuda_stream
        buf.mul_(2.0)
        _lib.check(L.pgcn_allreduce_sum_f32(comm, buf.data_ptr(), buf.numel(), s), "allreduce (captured)")
        _lib.check(L.pgcn_exchange_alltoallv_f32(comm, None, off, None, off, 128, s), "alltoallv (captured)")
        out.copy_(buf)
    torch.cuda.synchronize(dev)
    want = torch.arange(4096, dtype=torch.float32)
    for k in range(1, 4):
        g.replay()
        torch.cuda.synchronize(dev)
        assert torch.equal(out.cpu(), want * 2.0 ** k)      # a one-rank sum is the identity; the replays really ran
    _lib.check(L.pgcn_comm_destroy(comm), "comm_destroy")


@pytest.mark.parametrize("mtx,pv,P,L,f", [("karate.A.mtx", "karate.mtx.3.hp", 3, 3, 16),
                                          ("gemat11p.A.mtx", "gemat11.mtx.2.rp", 2, 3, 32)])
def test_run_multi_rank_on_one_gpu(dev, mtx, pv, P, L, f):
    """N > 1 with the real kernels: P processes on the one GPU, gloo transport (host-staged),
    comm-stream overlap on.  Checked against the oracle's restatement of run()."""
    import torch.multiprocessing as mp
    import torch.nn as nn
    import _workers
    seed = 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_workers.run_worker_gpu, args=(r, P, port, gpath(mtx), gpath(pv), L, f, seed, q))
             for r in range(P)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(P)], key=lambda r: r["rank"])
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    A = sp.csr_matrix(mmread(gpath(mtx))).astype(np.float32)
    n = A.shape[0]
    part = read_partvec(gpath(pv))
    torch.manual_seed(seed)
    w0 = [nn.Linear(f, f, bias=False).weight.detach().numpy() for _ in range(L)]
    H0 = np.repeat(np.arange(n, dtype=np.float32)[:, None], f, axis=1)
    losses, Ws = oracle.pgcn_train_np(A, part, P, w0, H0, np.arange(n) % f, epochs=5)
    printed = [float(x) for x in re.findall(r"Epoch \d{5} \| Loss ([0-9.]+)", res[0]["stdout"])]
    np.testing.assert_allclose(printed, losses[1:], rtol=1e-4, atol=1e-4)
    for r in res:
        assert r["provider"] == "HipKernels" and r["overlap"]
        for i, w in enumerate(r["weights"]):
            assert rel_err(w, Ws[i]) < 5e-4
    rows = sum(v.size for r in range(P) for v in oracle.communication_maps(A, part, r, P)[0].values())
    m = re.search(r"total_vol: (\d+) total_nmsg: (\d+)", res[0]["stdout"])
    assert int(m.group(1)) == rows * 5 * L * 2 and int(m.group(2)) == P * (P - 1) * 5 * L * 2


def test_pargcn_cli_on_reference_inputs(dev, tmp_path):
    """`pargcn.py -p DIR -c CONFIG` on a directory written by the reference's GCN-HP tool, real kernels."""
    import tarfile
    from conftest import GOLDEN
    pargcn, io_ = pkg("pargcn"), pkg("pargcn_io")
    with tarfile.open(os.path.join(GOLDEN, "pargcn", "gemat11p_k3.tar.gz")) as tf:
        tf.extractall(tmp_path)
    d_ = str(tmp_path / "out_gemat11p_k3")
    os.environ["PGCN_SEED"] = "5"
    buf = io.StringIO()
    errs, Wn, Hout, part = pargcn.main(["-p", d_, "-c", os.path.join(d_, "config")], out=buf)
    prob = io_.load_directory(d_)
    d = prob["d"]
    n = d[0]
    # HB/gemat11 is unsymmetric: the engine, like the reference, multiplies by what the conn files deliver
    A, dropped = oracle.drop_undelivered(prob["A"], prob["part"], prob["conn"], prob["k"])
    assert dropped > 0
    err, Wc, Hl, _ = oracle.pargcn_train(A, [0] * n, 1, d, pargcn.init_weights(d, 5),
                                         np.ones((n, d[1]), np.float32), prob["Y"], prob["Ymask"])
    np.testing.assert_allclose(errs, err, rtol=1e-5)
    for l in Wc:
        assert rel_err(Wn[l].cpu().numpy(), Wc[l]) < TOL
    assert rel_err(Hout.cpu().numpy(), Hl[part.owned.numpy()]) < TOL
    assert buf.getvalue().startswith("nlayers:3") and "time :" in buf.getvalue()


@pytest.mark.parametrize("name,P", [("ref_minibatch_karateA", 1), ("ref_minibatch_gemat11pA", 1), ("ref_minibatch_gemat11pA", 2)])
def test_minibatch_driver_real_kernels(dev, name, P, tmp_path):
    """PGCN_minibatch.run() with the HIP kernels vs the reference's PGCN-Mini-batch.py (P=1 golden)."""
    import json, pickle
    import torch.multiprocessing as mp
    import _workers
    from conftest import GOLDEN
    arrays = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
    pv = gpath(name + ".partvec.pickle")
    if P > 1:
        pv = str(tmp_path / "pv.pickle")
        n = len(pickle.load(open(gpath(name + ".partvec.pickle"), "rb")))
        with open(pv, "wb") as f:
            pickle.dump([int(x) for x in np.random.default_rng(0).integers(0, P, n)], f)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_workers.minibatch_worker,
                         args=(r, P, port, gpath(meta["mtx"]), pv, meta["f"], meta["batch_size"], meta["seed"], True, q))
             for r in range(P)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(P)], key=lambda r: r["rank"])
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    got = [float(x) for x in re.findall(r"Epoch \d{5} \| Loss ([0-9.]+)", res[0]["stdout"])]
    if P == 1:
        np.testing.assert_allclose(got, meta["losses"], rtol=5e-5, atol=2e-4)
        for i, w in enumerate(res[0]["weights"]):
            assert rel_err(w, arrays["w1_%d" % i]) < 5e-4
    else:   # replicas stay identical and training makes progress; losses differ by the log(f) constant
        assert got[-1] < got[0]
        for a, b in zip(res[0]["weights"], res[1]["weights"]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n,f,ld", [(1, 1, 1), (513, 16, 16), (4000, 128, 128), (777, 100, 104), (300, 1024, 1024), (50, 65, 65), (1001, 256, 260),
                                     (35, 64, 64), (18, 4, 4)])
def test_row_nll_kernels_vs_torch(K, dev, n, f, ld):
    """pgcn_nll_rows_f32 / _backward_f32 == F.nll_loss(F.log_softmax(x), y, reduction='sum') and its gradient
    (GPU/PGCN.py:214-215), incl. rows with -inf entries and large magnitudes."""
    import torch.nn.functional as F
    P = pkg("PGCN")
    g = torch.Generator(device=dev); g.manual_seed(n + f)
    buf = torch.randn((n, ld), device=dev, generator=g) * 8
    if n > 3 and f > 2:
        buf[1, 0] = float("-inf")
        buf[2, :f] = 300.0
        buf[3, 1] = -300.0
    x = buf[:, :f].detach().requires_grad_(True)
    y = torch.randint(0, f, (n,), device=dev, generator=g)
    if n > 3 and f > 2:
        y[1] = 1
    out = K.nll_rows(x.detach(), y)
    assert out is not None
    ref_rows = F.nll_loss(F.log_softmax(x.detach().double(), 1), y, reduction="none")
    assert float((out[0].double() - ref_rows).abs().max()) <= 1e-5 * max(1.0, float(ref_rows.abs().max()))
    loss = P._RowNLLSum.apply(x, y, K) / 7.0
    loss.backward()
    xr = x.detach().double().requires_grad_(True)
    (F.nll_loss(F.log_softmax(xr, 1), y, reduction="sum") / 7.0).backward()
    assert abs(float(loss) - float(ref_rows.sum() / 7.0)) <= 1e-5 * max(1.0, abs(float(ref_rows.sum() / 7.0)))
    assert float((x.grad.double() - xr.grad).abs().max()) <= 2e-6
    assert K.nll_rows(torch.zeros((4, 1025), device=dev), torch.zeros(4, dtype=torch.int64, device=dev)) is None
