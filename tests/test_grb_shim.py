"""The stand-ins that let the reference's Parallel-GCN/main.c compile here (oracle/shim/, test infrastructure) are
themselves checked against independent implementations: the GraphBLAS subset against scipy.sparse on random matrices
(patterns exactly, values to fp32 round-off) plus the specification's corner cases main.c leans on (union vs
intersection, replace without a mask, accumulate over the union, duplicates in build, setElement overwrite, error
codes), the MPI subset by a known-answer program over 1..4 forked ranks (oracle/shim/mpi_selftest.c)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT

BUILD = os.path.join(ROOT, "oracle", "_build")
u64p, f32p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float)
UNARY = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)
BINARY = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
SUCCESS, DIMENSION_MISMATCH, OUTPUT_NOT_EMPTY, INSUFFICIENT_SPACE, INDEX_OUT_OF_BOUNDS, INVALID_VALUE = 0, 8, 9, 11, 12, 5


@pytest.fixture(scope="module")
def G():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "shim"], check=True)
    L = ctypes.CDLL(os.path.join(BUILD, "libgrbshim.so"))
    for name in ("GrB_FP32", "GrB_PLUS_FP32", "GrB_MINUS_FP32", "GrB_TIMES_FP32", "GrB_DIV_FP32", "GxB_PLUS_TIMES_FP32",
                 "GxB_PLUS_SECOND_FP32", "GrB_DESC_R", "GrB_DESC_T0", "GrB_DESC_RT1", "GrB_ALL"):
        setattr(L, "v_" + name, ctypes.c_void_p.in_dll(L, name))
    L.GrB_Matrix_new.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
    L.GrB_Matrix_build_FP32.argtypes = [ctypes.c_void_p, u64p, u64p, f32p, ctypes.c_uint64, ctypes.c_void_p]
    L.GrB_Matrix_extractTuples_FP32.argtypes = [u64p, u64p, f32p, u64p, ctypes.c_void_p]
    L.GrB_Matrix_setElement_FP64.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_uint64, ctypes.c_uint64]
    L.GrB_Matrix_nvals.argtypes = [u64p, ctypes.c_void_p]
    L.GrB_Matrix_clear.argtypes = [ctypes.c_void_p]
    L.GrB_Matrix_free.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    L.GrB_mxm.argtypes = [ctypes.c_void_p] * 7
    L.GrB_Matrix_eWiseAdd_BinaryOp.argtypes = [ctypes.c_void_p] * 7
    L.GrB_Matrix_eWiseMult_BinaryOp.argtypes = [ctypes.c_void_p] * 7
    L.GrB_Matrix_apply.argtypes = [ctypes.c_void_p] * 6
    L.GrB_Matrix_apply_BinaryOp2nd_FP32.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_float, ctypes.c_void_p]
    L.GrB_Matrix_assign_FP32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                                                ctypes.c_uint64, ctypes.c_void_p]
    L.GrB_Matrix_reduce_FP32.argtypes = [f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.GrB_UnaryOp_new.argtypes = [ctypes.POINTER(ctypes.c_void_p), UNARY, ctypes.c_void_p, ctypes.c_void_p]
    L.GrB_BinaryOp_new.argtypes = [ctypes.POINTER(ctypes.c_void_p), BINARY, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.GrB_Monoid_new_FP32.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_float]
    return L


def new(G, shape):
    m = ctypes.c_void_p()
    assert G.GrB_Matrix_new(ctypes.byref(m), G.v_GrB_FP32, shape[0], shape[1]) == SUCCESS
    return m


def put(G, M: sp.spmatrix, shuffle=None):
    """A shim matrix holding the stored entries of M (explicit zeros included)."""
    M = sp.coo_matrix(M)
    m = new(G, M.shape)
    order = np.arange(M.nnz) if shuffle is None else shuffle.permutation(M.nnz)
    I, J, X = (np.ascontiguousarray(M.row[order], np.uint64), np.ascontiguousarray(M.col[order], np.uint64),
               np.ascontiguousarray(M.data[order], np.float32))
    assert G.GrB_Matrix_build_FP32(m, I.ctypes.data_as(u64p), J.ctypes.data_as(u64p), X.ctypes.data_as(f32p), M.nnz,
                                   G.v_GrB_PLUS_FP32) == SUCCESS
    return m


def get(G, m, shape):
    """(rows, cols, values) in the order extractTuples returns them."""
    n = ctypes.c_uint64()
    assert G.GrB_Matrix_nvals(ctypes.byref(n), m) == SUCCESS
    I, J, X = np.zeros(n.value, np.uint64), np.zeros(n.value, np.uint64), np.zeros(n.value, np.float32)
    cap = ctypes.c_uint64(n.value)
    assert G.GrB_Matrix_extractTuples_FP32(I.ctypes.data_as(u64p), J.ctypes.data_as(u64p), X.ctypes.data_as(f32p),
                                           ctypes.byref(cap), m) == SUCCESS
    assert cap.value == n.value
    if n.value:
        key = I.astype(np.int64) * shape[1] + J.astype(np.int64)
        assert (np.diff(key) > 0).all(), "tuples must come out row-major, without duplicates"
    return I.astype(np.int64), J.astype(np.int64), X


def dense_and_mask(I, J, X, shape):
    D, M = np.zeros(shape, np.float64), np.zeros(shape, bool)
    D[I, J] = X
    M[I, J] = True
    return D, M


def rand(rng, shape, density):
    M = sp.random(shape[0], shape[1], density=density, random_state=rng, data_rvs=lambda k: rng.uniform(-1, 1, k)).astype(np.float32)
    return sp.coo_matrix(M)


def _pattern(M):
    M = sp.coo_matrix(M)
    P = np.zeros(M.shape, bool)
    P[M.row, M.col] = True
    return P


@pytest.mark.parametrize("seed,ta,tb,second,accum", [(0, 0, 0, 0, 0), (1, 0, 0, 0, 1), (2, 1, 0, 0, 1), (3, 0, 1, 0, 0),
                                                     (4, 0, 0, 1, 0), (5, 0, 0, 1, 1)])
def test_mxm_against_scipy(G, seed, ta, tb, second, accum):
    """C<> (+)= A' (+.x) B' for the descriptor / semiring / accumulator forms of main.c:250,271,295,303,407,415."""
    if ta and tb:
        pytest.skip("main.c never transposes both")
    rng = np.random.default_rng(seed)
    m, k, n = 37, 23, 19
    A = rand(rng, (k, m) if ta else (m, k), 0.15)
    B = rand(rng, (n, k) if tb else (k, n), 0.2)
    C0 = rand(rng, (m, n), 0.1)
    a, b, c = put(G, A, rng), put(G, B, rng), put(G, C0)
    desc = G.v_GrB_DESC_T0 if ta else (G.v_GrB_DESC_RT1 if tb else None)
    rc = G.GrB_mxm(c, None, G.v_GrB_PLUS_FP32 if accum else None, G.v_GxB_PLUS_SECOND_FP32 if second else G.v_GxB_PLUS_TIMES_FP32,
                   a, b, desc)
    assert rc == SUCCESS
    Ae, Be = (A.T if ta else A).tocsr().astype(np.float64), (B.T if tb else B).tocsr().astype(np.float64)
    if second:                                            # SECOND(a, b) = b: A contributes its pattern only
        Ae = sp.csr_matrix((np.ones(Ae.nnz), Ae.indices, Ae.indptr), shape=Ae.shape)
    T = (Ae @ Be).toarray()
    Tp = (_pattern(Ae).astype(np.int64) @ _pattern(Be).astype(np.int64)) > 0
    C0d, C0p = dense_and_mask(C0.row, C0.col, C0.data, (m, n))
    want, wantp = (C0d + T, C0p | Tp) if accum else (T, Tp)      # accumulate over the UNION; no accumulator: C is replaced
    got, gotp = dense_and_mask(*get(G, c, (m, n)), (m, n))
    assert np.array_equal(gotp, wantp)
    np.testing.assert_allclose(got[gotp], want[wantp], rtol=2e-6, atol=2e-7)
    # dimension check and the refusal of masks (main.c passes none)
    assert G.GrB_mxm(new(G, (m + 1, n)), None, None, G.v_GxB_PLUS_TIMES_FP32, a, b, desc) == DIMENSION_MISMATCH
    assert G.GrB_mxm(c, c, None, G.v_GxB_PLUS_TIMES_FP32, a, b, desc) == INVALID_VALUE


def test_mxm_sums_in_ascending_k_and_keeps_explicit_zeros(G):
    A = sp.coo_matrix((np.array([1e8, 1.0, -1e8], np.float32), ([0, 0, 0], [0, 1, 2])), shape=(1, 3))
    B = sp.coo_matrix((np.ones(3, np.float32), ([0, 1, 2], [0, 0, 0])), shape=(3, 1))
    c = new(G, (1, 1))
    assert G.GrB_mxm(c, None, None, G.v_GxB_PLUS_TIMES_FP32, put(G, A), put(G, B), None) == SUCCESS
    I, J, X = get(G, c, (1, 1))
    assert X.tolist() == [0.0]              # (1e8 + 1) - 1e8 in fp32, k ascending; the entry stays although it is zero


def test_elementwise_union_and_intersection(G):
    """eWiseAdd applies the operator only where BOTH exist and copies the lone entries (the loss of main.c:318 and the
    MINUS of :327 rely on it: a lone Y entry would come through as +y, not -y); eWiseMult keeps the intersection."""
    rng = np.random.default_rng(7)
    shape = (29, 17)
    A, B = rand(rng, shape, 0.3), rand(rng, shape, 0.3)
    a, b, c = put(G, A, rng), put(G, B, rng), new(G, shape)
    Ad, Ap = dense_and_mask(A.row, A.col, A.data, shape)
    Bd, Bp = dense_and_mask(B.row, B.col, B.data, shape)
    assert G.GrB_Matrix_eWiseAdd_BinaryOp(c, None, None, G.v_GrB_MINUS_FP32, a, b, G.v_GrB_DESC_R) == SUCCESS
    got, gotp = dense_and_mask(*get(G, c, shape), shape)
    assert np.array_equal(gotp, Ap | Bp)
    want = np.where(Ap & Bp, Ad.astype(np.float32) - Bd.astype(np.float32), np.where(Ap, Ad, Bd))
    assert np.array_equal(got[gotp].astype(np.float32), want[gotp].astype(np.float32))
    assert G.GrB_Matrix_eWiseMult_BinaryOp(c, None, None, G.v_GrB_DIV_FP32, a, b, G.v_GrB_DESC_R) == SUCCESS
    got, gotp = dense_and_mask(*get(G, c, shape), shape)
    assert np.array_equal(gotp, Ap & Bp)
    assert np.array_equal(got[gotp].astype(np.float32), (Ad.astype(np.float32) / np.where(Bp, Bd, 1).astype(np.float32))[gotp])
    # output aliasing an input (main.c:327-328: H = H - Y; H = H / T)
    assert G.GrB_Matrix_eWiseMult_BinaryOp(a, None, None, G.v_GrB_TIMES_FP32, a, b, G.v_GrB_DESC_R) == SUCCESS
    got, gotp = dense_and_mask(*get(G, a, shape), shape)
    assert np.array_equal(gotp, Ap & Bp)
    assert np.array_equal(got[gotp].astype(np.float32), (Ad.astype(np.float32) * Bd.astype(np.float32))[gotp])
    d = new(G, (3, 3))
    assert G.GrB_Matrix_eWiseAdd_BinaryOp(d, None, None, G.v_GrB_PLUS_FP32, a, b, None) == DIMENSION_MISMATCH


def test_apply_user_operators_assign_reduce(G):
    rng = np.random.default_rng(9)
    shape = (11, 13)
    A = rand(rng, shape, 0.4)
    a, c = put(G, A), new(G, shape)
    Ad, Ap = dense_and_mask(A.row, A.col, A.data, shape)

    @UNARY
    def square_plus_one(z, x):
        ctypes.cast(z, f32p)[0] = ctypes.cast(x, f32p)[0] ** 2 + 1

    @BINARY
    def update(z, x, y):                               # main.c:75-77 with alpha = 0.5
        ctypes.cast(z, f32p)[0] = ctypes.cast(x, f32p)[0] - 0.5 * ctypes.cast(y, f32p)[0]

    uop, bop, mon = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert G.GrB_UnaryOp_new(ctypes.byref(uop), square_plus_one, G.v_GrB_FP32, G.v_GrB_FP32) == SUCCESS
    assert G.GrB_BinaryOp_new(ctypes.byref(bop), update, G.v_GrB_FP32, G.v_GrB_FP32, G.v_GrB_FP32) == SUCCESS
    assert G.GrB_Matrix_apply(c, None, None, uop, a, G.v_GrB_DESC_R) == SUCCESS
    got, gotp = dense_and_mask(*get(G, c, shape), shape)
    assert np.array_equal(gotp, Ap)                     # the pattern of A, also where f(a) could be anything
    np.testing.assert_allclose(got[gotp], (Ad ** 2 + 1)[Ap], rtol=1e-6)
    assert G.GrB_Matrix_apply_BinaryOp2nd_FP32(c, None, None, G.v_GrB_DIV_FP32, a, 34.0, G.v_GrB_DESC_R) == SUCCESS     # main.c:335
    got, gotp = dense_and_mask(*get(G, c, shape), shape)
    assert np.array_equal(got[gotp].astype(np.float32), (Ad.astype(np.float32) / np.float32(34))[Ap])
    # C(:,:) = 0 makes C dense (main.c:414), and the user operator then runs on every entry (main.c:430)
    w = new(G, shape)
    assert G.GrB_Matrix_assign_FP32(w, None, None, 0.0, G.v_GrB_ALL, 0, G.v_GrB_ALL, 0, None) == SUCCESS
    got, gotp = dense_and_mask(*get(G, w, shape), shape)
    assert gotp.all() and not got.any()
    assert G.GrB_Matrix_eWiseAdd_BinaryOp(w, None, None, bop, w, a, G.v_GrB_DESC_R) == SUCCESS
    got, gotp = dense_and_mask(*get(G, w, shape), shape)
    assert gotp.all()
    assert np.array_equal(got.astype(np.float32), np.where(Ap, (0 - 0.5 * Ad).astype(np.float32), 0).astype(np.float32))
    assert G.GrB_Matrix_assign_FP32(w, None, None, 1.0, None, 0, G.v_GrB_ALL, 0, None) == INVALID_VALUE     # only (:,:)
    # reduce: the plain fp32 running sum in (i, j) order; the monoid's "identity" (main.c:218 passes 1.0) only for an empty matrix
    assert G.GrB_Monoid_new_FP32(ctypes.byref(mon), G.v_GrB_PLUS_FP32, 1.0) == SUCCESS
    s = ctypes.c_float(-5)
    assert G.GrB_Matrix_reduce_FP32(ctypes.byref(s), None, mon, a, None) == SUCCESS
    I, J, X = get(G, a, shape)
    run = np.float32(X[0])
    for v in X[1:]:
        run = np.float32(run + v)
    assert s.value == run
    assert G.GrB_Matrix_reduce_FP32(ctypes.byref(s), None, mon, new(G, shape), None) == SUCCESS and s.value == 1.0


def test_build_setelement_extract_errors(G):
    shape = (5, 4)
    m = new(G, shape)
    I = np.array([3, 0, 3, 0, 3], np.uint64); J = np.array([1, 2, 1, 0, 1], np.uint64); X = np.array([1, 2, 4, 8, 16], np.float32)
    args = (I.ctypes.data_as(u64p), J.ctypes.data_as(u64p), X.ctypes.data_as(f32p), 5)
    assert G.GrB_Matrix_build_FP32(m, *args, G.v_GrB_PLUS_FP32) == SUCCESS
    r, c, x = get(G, m, shape)
    assert (r.tolist(), c.tolist(), x.tolist()) == ([0, 0, 3], [0, 2, 1], [8.0, 2.0, 21.0])      # duplicates folded by dup
    assert G.GrB_Matrix_build_FP32(m, *args, G.v_GrB_PLUS_FP32) == OUTPUT_NOT_EMPTY                 # main.c clears first (:292)
    assert G.GrB_Matrix_clear(m) == SUCCESS
    r, c, x = get(G, m, shape)
    assert r.size == 0
    bad = np.array([5], np.uint64)
    assert G.GrB_Matrix_build_FP32(m, bad.ctypes.data_as(u64p), J.ctypes.data_as(u64p), X.ctypes.data_as(f32p), 1,
                                   G.v_GrB_PLUS_FP32) == INDEX_OUT_OF_BOUNDS
    # setElement: pending tuples, a later one overwrites (the int literal of main.c:543 arrives as 1.0)
    assert G.GrB_Matrix_setElement_FP64(m, 1, 2, 2) == SUCCESS
    assert G.GrB_Matrix_setElement_FP64(m, 0.5, 0, 3) == SUCCESS
    assert G.GrB_Matrix_setElement_FP64(m, 7.25, 2, 2) == SUCCESS
    assert G.GrB_Matrix_setElement_FP64(m, 1.0, 5, 0) != SUCCESS
    r, c, x = get(G, m, shape)
    assert (r.tolist(), c.tolist(), x.tolist()) == ([0, 2], [3, 2], [0.5, 7.25])
    n = ctypes.c_uint64(1)
    assert G.GrB_Matrix_extractTuples_FP32(I.ctypes.data_as(u64p), J.ctypes.data_as(u64p), X.ctypes.data_as(f32p),
                                           ctypes.byref(n), m) == INSUFFICIENT_SPACE
    h = ctypes.c_void_p(m.value)
    assert G.GrB_Matrix_free(ctypes.byref(h)) == SUCCESS and h.value is None


@pytest.mark.parametrize("np_", [1, 2, 3, 4])
def test_mpi_stand_in_known_answers(G, np_):
    out = subprocess.run([os.path.join(BUILD, "mpi_selftest")], env=dict(os.environ, MPISHIM_NP=str(np_), MPISHIM_SEED="1234"),
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "mpi selftest ok %d" % np_
