"""N1 (fast ingest): the C++ MatrixMarket reader returns exactly what scipy.io.mmread returns
(the loader the reference uses, GPU/PGCN.py:171) -- same entries, bit-identical fp32 values."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.io import mmread as sp_mmread
from scipy.io import mmwrite

import torch

from conftest import gpath, pkg


def _canon(A):
    A = sp.coo_matrix(A)
    v = A.data.astype(np.float32)
    order = np.lexsort((v, A.col, A.row))
    return A.shape, A.row[order].astype(np.int64), A.col[order].astype(np.int64), v[order]


def _same(a, b):
    sa, ra, ca, va = _canon(a)
    sb, rb, cb, vb = _canon(b)
    assert sa == sb
    np.testing.assert_array_equal(ra, rb)
    np.testing.assert_array_equal(ca, cb)
    np.testing.assert_array_equal(va.view(np.uint32), vb.view(np.uint32))   # bit-exact


@pytest.mark.parametrize("name", ["karate.mtx", "karate.A.mtx", "gemat11.mtx", "gemat11p.mtx", "gemat11p.A.mtx"])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_matches_scipy_on_fixtures(name, threads):
    ingest = pkg("ingest")
    _same(ingest.mmread(gpath(name), nthreads=threads), sp_mmread(gpath(name)))


def test_flavours_and_formats(tmp_path):
    ingest = pkg("ingest")
    rng = np.random.default_rng(0)
    n = 300
    M = sp.random(n, n, density=0.05, random_state=1, format="coo", dtype=np.float64)
    M.data = (rng.standard_normal(M.nnz) * 10.0 ** rng.integers(-30, 30, M.nnz))
    cases = {}
    cases["real_general"] = (M, {})
    cases["real_prec17"] = (M, {"precision": 17})
    S = sp.coo_matrix(M + M.T)
    cases["real_symmetric"] = (S, {"symmetry": "symmetric"})
    K = sp.coo_matrix(M - M.T)
    cases["skew"] = (K, {"symmetry": "skew-symmetric"})
    cases["integer"] = (sp.coo_matrix((rng.integers(-50, 50, M.nnz), (M.row, M.col)), shape=M.shape), {"field": "integer"})
    cases["pattern_sym"] = (sp.coo_matrix((np.ones(S.nnz), (S.row, S.col)), shape=S.shape), {"field": "pattern", "symmetry": "symmetric"})
    cases["rect"] = (sp.random(50, 700, density=0.1, random_state=2, format="coo"), {})
    for nm, (A, kw) in cases.items():
        p = str(tmp_path / (nm + ".mtx"))
        mmwrite(p, A, comment="written by the test\n second comment line", **kw)
        for t in (1, 4):
            _same(ingest.mmread(p, nthreads=t), sp_mmread(p))
        info = ingest.mtx_info(p)
        assert (info["nrows"], info["ncols"]) == A.shape
    # hand-written oddities: blank lines, tabs, CRLF, D exponents, '+' signs, trailing spaces
    p = str(tmp_path / "odd.mtx")
    with open(p, "w", newline="") as f:
        f.write("%%MatrixMarket MATRIX Coordinate Real General\r\n% c\r\n\r\n  3 4\t5  \r\n"
                "1 1 +1.5\r\n2\t3   -2.5e-3 \r\n\r\n3 4 1D2\r\n3 1 .5\r\n1 4 7.\r\n")
    got = ingest.mmread(p)
    ref = sp.coo_matrix((np.array([1.5, -2.5e-3, 100.0, 0.5, 7.0]), ([0, 1, 2, 2, 0], [0, 2, 3, 0, 3])), shape=(3, 4))
    _same(got, ref)


def test_errors(tmp_path):
    ingest, _lib = pkg("ingest"), pkg("_lib")
    with pytest.raises(_lib.PgcnError):
        ingest.mmread(str(tmp_path / "missing.mtx"))
    p = str(tmp_path / "short.mtx")
    open(p, "w").write("%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1.0\n2 2 2.0\n")
    with pytest.raises(_lib.PgcnError, match="entry lines"):
        ingest.mmread(p)
    p = str(tmp_path / "range.mtx")
    open(p, "w").write("%%MatrixMarket matrix coordinate real general\n3 3 1\n4 1 1.0\n")
    with pytest.raises(_lib.PgcnError, match="out of range"):
        ingest.mmread(p)
    # array format is handed to scipy
    p = str(tmp_path / "dense.mtx")
    mmwrite(p, np.arange(6.0).reshape(2, 3))
    _same(ingest.mmread(p), sp.coo_matrix(sp_mmread(p)))


# ---- native comm-map builder / partition loader against the reference's own maps --------------
from conftest import SPMM_CASES, golden, read_partvec  # noqa: E402


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES)
def test_native_comm_maps_match_reference(name, mtx, pv, P, threads):
    """pgcn_build_comm_maps == compute_communication_maps of GPU/PGCN.py:37-51 (golden files made
    by running the reference) -- integer work, exact."""
    ingest = pkg("ingest")
    arrays, meta = golden(name)
    A = sp.coo_matrix(sp_mmread(gpath(mtx)))
    part = np.array(read_partvec(gpath(pv)), dtype=np.int32)
    for r in range(P):
        send, recv = ingest.comm_maps(A.row, A.col, part, r, P, nthreads=threads)
        assert sorted(send) == sorted(recv) == [q for q in range(P) if q != r]
        for q in send:
            np.testing.assert_array_equal(send[q], arrays["send_%d_%d" % (r, q)])
            np.testing.assert_array_equal(recv[q], arrays["recv_%d_%d" % (r, q)])
        # entries touching rank r are enough (what a rank holding only its rows + columns passes)
        touch = (part[A.row] == r) | (part[A.col] == r)
        s2, r2 = ingest.comm_maps(A.row[touch], A.col[touch], part, r, P, nthreads=threads)
        for q in send:
            np.testing.assert_array_equal(s2[q], send[q])
            np.testing.assert_array_equal(r2[q], recv[q])


def test_native_comm_maps_large_random_vs_sets():
    ingest = pkg("ingest")
    rng = np.random.default_rng(5)
    n, nnz, P = 20000, 600000, 5                 # several 64-bit words per bit-map row, 4+ threads
    row, col = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    part = rng.integers(0, P, n).astype(np.int32)
    for r in (0, 3):
        send, recv = ingest.comm_maps(row, col, part, r, P)
        pr, pc = part[row], part[col]
        for q in range(P):
            if q == r:
                continue
            np.testing.assert_array_equal(recv[q], np.unique(col[(pr == r) & (pc == q)]))
            np.testing.assert_array_equal(send[q], np.unique(col[(pc == r) & (pr == q)]))


def test_native_comm_maps_edge_cases_and_errors():
    ingest, _lib = pkg("ingest"), pkg("_lib")
    e = np.zeros(0, dtype=np.int64)
    send, recv = ingest.comm_maps(e, e, np.zeros(4, np.int32), 0, 1)       # P = 1: no peers
    assert send == {} and recv == {}
    send, recv = ingest.comm_maps(e, e, np.array([0, 1, 1], np.int32), 1, 2)  # no entries
    assert send[0].size == 0 and recv[0].size == 0
    with pytest.raises(_lib.PgcnError):
        ingest.comm_maps(np.array([0]), np.array([7]), np.array([0, 1], np.int32), 0, 2)   # index out of range
    with pytest.raises(_lib.PgcnError):
        ingest.comm_maps(np.array([0]), np.array([1]), np.array([0, 2], np.int32), 0, 2)   # part id out of range
    with pytest.raises(_lib.PgcnError):
        ingest.comm_maps(np.array([0]), np.array([1]), np.array([0, 1], np.int32), 2, 2)   # rank out of range


@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES[:1] + SPMM_CASES[4:5] + SPMM_CASES[7:8])
def test_native_partition_loader(name, mtx, pv, P):
    """pgcn_load_mtx_partition == mmread + the np.in1d row mask of GPU/PGCN.py:53-64."""
    ingest, _lib = pkg("ingest"), pkg("_lib")
    _, meta = golden(name)
    A = sp.coo_matrix(sp_mmread(gpath(mtx)))
    part = np.array(read_partvec(gpath(pv)), dtype=np.int32)
    for r in range(P):
        got = ingest.load_partition(gpath(mtx), part, r)
        keep = part[A.row] == r
        _same(got, sp.coo_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=A.shape))
        assert got.nnz == meta["ranks"][r]["nnz_local"]
    with pytest.raises(_lib.PgcnError):
        ingest.load_partition(gpath(mtx), part[:-1], 0)                   # part vector length mismatch


def test_binary_csr_shards_round_trip(tmp_path):
    """pgcn_shard_write / _info / _read: one file per rank with only its rows, int64 row pointers, validated."""
    import scipy.sparse as sp
    ingest, _lib = pkg("ingest"), pkg("_lib")
    rng = np.random.default_rng(0)
    A = sp.random(300, 300, 0.05, random_state=2, dtype=np.float32, format="csr")
    A.sort_indices()
    part = rng.integers(0, 4, 300)
    paths = ingest.write_shards(str(tmp_path / "g"), A, part, 4)
    tot = 0
    for p, path in enumerate(paths):
        info = ingest.shard_info(path)
        own = np.nonzero(part == p)[0]
        assert (info["n"], info["rank"], info["nparts"], info["nrows"]) == (300, p, 4, own.size)
        sh = ingest.read_shard(path)
        assert np.array_equal(sh["rows"], own) and sh["rowptr"].dtype == np.int64 and sh["col"].dtype == np.int32
        r, c, v = ingest.shard_coo(sh)
        B = sp.csr_matrix((v, (r, c)), shape=A.shape)
        assert (abs(B - sp.csr_matrix(A.multiply((part == p)[:, None]))) > 0).nnz == 0
        tot += info["nnz"]
        assert os.path.getsize(path) == 64 + 8 * own.size + 8 * (own.size + 1) + 4 * (info["nnz"] + info["nnz"] % 2) + 4 * info["nnz"]
    assert tot == A.nnz
    # an empty part, bad inputs, a truncated file
    ingest.write_shard(str(tmp_path / "e.0.pgcsr"), 10, 0, 1, [], [0], [], [])
    assert ingest.read_shard(str(tmp_path / "e.0.pgcsr"))["nnz"] == 0
    with pytest.raises(_lib.PgcnError):
        ingest.write_shard(str(tmp_path / "bad.pgcsr"), 10, 0, 1, [3, 2], [0, 1, 2], [1, 1], [1.0, 1.0])     # rows not ascending
    with pytest.raises(_lib.PgcnError):
        ingest.write_shard(str(tmp_path / "bad.pgcsr"), 10, 2, 2, [1], [0, 0], [], [])                        # rank >= nparts
    data = open(paths[0], "rb").read()
    open(tmp_path / "cut.pgcsr", "wb").write(data[:len(data) // 2])
    with pytest.raises(_lib.PgcnError):
        ingest.read_shard(str(tmp_path / "cut.pgcsr"))
    open(tmp_path / "junk.pgcsr", "wb").write(b"x" * 100)
    with pytest.raises(_lib.PgcnError):
        ingest.shard_info(str(tmp_path / "junk.pgcsr"))


def test_plan_host_keeps_64_bit_entry_offsets():
    """papers100M-scale blocks hold more than 2^31 stored entries per rank: the plan's task records carry the
    absolute entry offset as two 32-bit words.  A row block whose row pointers START beyond 2^33 (the stride
    trick: no 8 G-entry arrays needed for the host plan) must come back with exact 64-bit offsets."""
    kernels = pkg("kernels")
    base = (1 << 33) + 12345
    lens = np.array([5, 0, 3000, 17, 1, 2500], dtype=np.int64)
    rowptr = base + np.concatenate([[0], np.cumsum(lens)])
    tasks, fix, nslots, seg = kernels.build_plan(rowptr, chunk=1024, force=True)
    kbeg = (tasks[:, 1].astype(np.int64) << 32) | (tasks[:, 0].astype(np.int64) & 0xFFFFFFFF)
    assert kbeg.min() == base and (kbeg >= base).all() and kbeg.max() < base + lens.sum()
    cover = np.zeros(int(lens.sum()), dtype=np.int64)
    for k, ln in zip(kbeg, tasks[:, 2]):
        cover[k - base:k - base + ln] += 1
    assert (cover == 1).all()                                        # every stored entry in exactly one task
    assert int(tasks[:, 2].max()) <= 1024 and nslots == 3 + 3 and fix.shape[0] == 2      # the two long rows are split in 3


def test_one_rank_shard_with_degree_file_rebuilds_the_ranks_partition(tmp_path):
    """tools/make_shards.py --only-rank (BASELINE config 4 at full size on ONE rank: bench.py --emulate-rank r/P --shards):
    the global key set equals the union of the rank-local generators', and the partition built from the rank's shard +
    the degree side file (build_partition_local(emulate=...): no collectives, the peers' needs derived from the
    symmetric pattern) is the partition the global build gives that rank."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    synth, partition, ingest = pkg("synth"), pkg("partition"), pkg("ingest")
    prefix = str(tmp_path / "mid")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_shards.py"), "--workload", "mid", "--ranks", "4", "--only-rank", "1",
                    "--scale", "0.25", "--out", prefix], check=True, capture_output=True)
    meta = json.load(open(prefix + ".meta.json"))
    n, P, r = meta["n"], 4, 1
    pv = synth.block_partvec(n, P)
    allk = synth.rmat_all_keys(n, meta["pairs"], seed=0)
    assert allk.numel() == meta["nnz_global"]
    for q in range(P):
        assert torch.equal(synth.rmat_shard_keys(n, meta["pairs"], q, pv, seed=0), allk[pv[allk // n] == q])
    deg = torch.from_numpy(np.load(prefix + ".degree.npy").astype(np.int64))
    assert torch.equal(deg, torch.bincount(allk // n, minlength=n))
    sh = ingest.read_shard(ingest.shard_path(prefix, r))
    rr, cc, vv = ingest.shard_coo(sh)
    e = partition.build_partition_local(torch.from_numpy(rr), torch.from_numpy(cc), torch.from_numpy(vv), n, pv, r, P,
                                        emulate={"gdeg": 2 * deg, "nnz_global": meta["nnz_global"]})
    row, col, val = synth.shard_normalize(n, allk, deg)
    g = partition.build_partition(row, col, val, n, pv, r, P)
    for name in ("owned", "send_idx", "send_owner", "halo_owner", "halo_global", "send_global"):
        assert torch.equal(getattr(g, name), getattr(e, name)), name
    assert g.round_send_off == e.round_send_off and g.round_recv_off == e.round_recv_off and g.nnz_global == e.nnz_global
    for a, b in ((g.A_loc, e.A_loc), (g.A_halo[0], e.A_halo[0]), (g.A_halo[1], e.A_halo[1]), (g.A_loc_T, e.A_loc_T), (g.A_halo_T[0], e.A_halo_T[0])):
        ra, ca, va = partition.full_csr(a)
        rb, cb, vb = partition.full_csr(b)
        assert torch.equal(ra, rb) and torch.equal(ca, cb) and torch.equal(va, vb)
