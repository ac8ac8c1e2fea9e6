"""N1 (fast ingest): the C++ MatrixMarket reader returns exactly what scipy.io.mmread returns
(the loader the reference uses, GPU/PGCN.py:171) -- same entries, bit-identical fp32 values."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.io import mmread as sp_mmread
from scipy.io import mmwrite

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
