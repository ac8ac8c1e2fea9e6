"""Pin the CPU oracle against golden vectors produced by the REFERENCE's own code
(tests/golden/make_golden.py ran /root/reference/GPU/PGCN.py under gloo).

Tolerance: the reference accumulates in fp32 in an implementation-defined order
(torch.sparse.mm on an uncoalesced COO), so fp32 results agree to a few ulps of
the row sum; we require 1e-5 relative to the largest magnitude (north_star)."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.io import mmread

from conftest import (SPMM_CASES, TRAIN_CASES, golden, golden_inputs, gpath, read_partvec, rel_err)
from oracle import oracle

TOL = 1e-5


@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES)
def test_comm_maps_match_reference(name, mtx, pv, P):
    arrays, meta = golden(name)
    A = mmread(gpath(mtx))
    part = read_partvec(gpath(pv))
    for r in range(P):
        send, recv = oracle.communication_maps(A, part, r, P)
        assert sorted(send) == [q for q in range(P) if q != r]
        for q in send:
            np.testing.assert_array_equal(send[q], arrays["send_%d_%d" % (r, q)])
            np.testing.assert_array_equal(recv[q], arrays["recv_%d_%d" % (r, q)])
        m = meta["ranks"][r]
        assert m["n_local"] == int((np.asarray(part) == r).sum())
        # PGCN.py:105,113 count rows per message
        assert m["stats_fwd"]["send_volume"] == sum(v.size for v in send.values())
        assert m["stats_fwd"]["recv_volume"] == sum(v.size for v in recv.values())


@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES)
def test_aggregate_forward_matches_reference_pspmm(name, mtx, pv, P):
    arrays, meta = golden(name)
    A = sp.csr_matrix(mmread(gpath(mtx))).astype(np.float32)
    part = read_partvec(gpath(pv))
    H, _ = golden_inputs(A.shape[0], meta["f"], meta["seed"])
    got = oracle.dist_aggregate(A, part, P, H)            # C, fp32
    assert rel_err(got, arrays["fwd"]) < TOL
    got64, rows = oracle.dist_aggregate_messages(A, part, P, H)   # numpy, float64, explicit messages
    assert rel_err(got64, arrays["fwd"]) < TOL
    for r in range(P):
        assert rows[r].sum() == meta["ranks"][r]["stats_fwd"]["send_volume"]
    # plain SpMM (P = 1 semantics) agrees as well
    assert rel_err(oracle.spmm(A, H), arrays["fwd"]) < TOL


@pytest.mark.parametrize("name,mtx,pv,P", [c for c in SPMM_CASES if c[3] <= 2])
def test_aggregate_backward_matches_reference_pspmm(name, mtx, pv, P):
    arrays, meta = golden(name)
    A = sp.csr_matrix(mmread(gpath(mtx))).astype(np.float32)
    _, G = golden_inputs(A.shape[0], meta["f"], meta["seed"])
    At = sp.csr_matrix(A.T)
    assert rel_err(oracle.spmm(At, G), arrays["bwd"]) < TOL       # PGCN.py:132 A.t() @ grad


@pytest.mark.parametrize("name,mtx,pv", TRAIN_CASES)
def test_pgcn_training_matches_reference_run(name, mtx, pv):
    arrays, meta = golden(name)
    A = sp.csr_matrix(mmread(gpath(mtx))).astype(np.float32)
    n, f, L = A.shape[0], meta["f"], meta["nlayers"]
    H0 = np.repeat(np.arange(n, dtype=np.float32)[:, None], f, axis=1)   # PGCN.py:186-188
    labels = np.arange(n) % f                                            # PGCN.py:192
    w0 = [arrays["w0_%d" % i] for i in range(L)]
    losses, Ws = oracle.pgcn_train_np(A, [0] * n, 1, w0, H0, labels, epochs=5)
    np.testing.assert_allclose(losses, arrays["losses"], rtol=2e-5)
    for i in range(L):
        assert rel_err(Ws[i], arrays["w1_%d" % i]) < 1e-4


def test_gather_scatter():
    rng = np.random.default_rng(0)
    H = rng.random((50, 6), dtype=np.float32)
    idx = rng.permutation(50)[:17].astype(np.int32)
    np.testing.assert_array_equal(oracle.gather_rows(H, idx), H[idx])
    src = rng.random((17, 6), dtype=np.float32)
    X = H.copy()
    oracle.scatter_rows(X, idx, src, accumulate=False)
    ref = H.copy(); ref[idx] = src
    np.testing.assert_array_equal(X, ref)
    oracle.scatter_rows(X, idx, src, accumulate=True)
    ref[idx] += src
    np.testing.assert_array_equal(X, ref)


def test_normalisation_matches_reference_preprocess():
    # preprocess/GrB-GNN-IDG.py wrote *.A.mtx with 3 significant digits
    for raw, norm in (("karate.mtx", "karate.A.mtx"), ("gemat11p.mtx", "gemat11p.A.mtx")):
        mine = oracle.normalize_adjacency(mmread(gpath(raw))).toarray()
        ref = sp.csr_matrix(mmread(gpath(norm))).toarray()
        assert (mine != 0).sum() == (ref != 0).sum()
        np.testing.assert_allclose(mine, ref, rtol=6e-3)


def test_pargcn_c_vs_float64_shadow():
    """Parallel-GCN/main.c training loop: fp32 C restatement vs float64 numpy
    restatement (parity vs a GraphBLAS binary is UNPINNED: it cannot be built here)."""
    A = oracle.normalize_adjacency(mmread(gpath("gemat11p.mtx")))
    A = ((A + A.T) * 0.5).tocsr().astype(np.float32)   # main.c:376 relies on A = A^T
    n = A.shape[0]
    d = [n, 16, 16, 2]
    rng = np.random.default_rng(3)
    W = {l: (rng.random((d[l], d[l + 1]), dtype=np.float32) * 2 - 1) * np.sqrt(6.0 / (d[l] + d[l + 1]))
         for l in (1, 2)}
    H0 = np.ones((n, 16), np.float32)                       # main.c:650-685
    Y = np.zeros((n, 2), np.float32); Y[:, 1] = 1           # GrB-GNN-IDG.py:76-78
    Ymask = np.zeros((n, 2), np.uint8); Ymask[:, 1] = 1
    part = read_partvec(gpath("gemat11.mtx.3.hp"))
    err1, W1, H1, st1 = oracle.pargcn_train(A, [0] * n, 1, d, W, H0, Y, Ymask)
    err3, W3, H3, st3 = oracle.pargcn_train(A, part, 3, d, W, H0, Y, Ymask)
    errd, Wd, Hd = oracle.pargcn_train_np(A, d, W, H0, Y, Ymask)
    np.testing.assert_allclose(err1, errd, rtol=1e-5)
    np.testing.assert_allclose(err3, errd, rtol=1e-5)
    assert errd[2] < errd[0]
    for l in (1, 2):
        assert rel_err(W1[l], Wd[l]) < TOL and rel_err(W3[l], Wd[l]) < TOL
    assert rel_err(H3, Hd) < TOL
    # volume (main.c:264 counts scalars): rows per exchange x widths; per epoch 2 forward
    # exchanges (widths 16,16) + 2 backward exchanges (widths 2,16)
    rows = sum(v.size for r in range(3) for v in oracle.communication_maps(A, part, r, 3)[0].values())
    assert st1.sum() == 0
    assert st3[:, 0].sum() == rows * (16 + 16 + 2 + 16) * 3
    assert st3[:, 1].sum() == 6 * 4 * 3


def test_fast_fp32_epoch_is_the_float64_shadow():
    """oracle.pgcn_epochs_f32 (bench.py's cpu_baseline leg: the GPU line's own epoch on the host cores, fp32, OpenMP SpMM +
    BLAS GEMMs) against pgcn_train_np, the float64 shadow the GPU tests hold run() to."""
    import scipy.sparse as sp
    from scipy.io import mmread
    from conftest import gpath, rel_err
    A = sp.csr_matrix(mmread(gpath("cora.A.mtx"))).astype(np.float32)
    n, f = A.shape[0], 16
    rng = np.random.default_rng(0)
    w0 = [(rng.random((f, f), dtype=np.float32) - 0.5) * 0.5 for _ in range(3)]
    H0 = np.repeat(np.arange(n, dtype=np.float32)[:, None], f, axis=1) / n
    lab = np.arange(n) % f
    l64, _ = oracle.pgcn_train_np(A, [0] * n, 1, w0, H0, lab, epochs=4)
    l32, secs = oracle.pgcn_epochs_f32(oracle._csr_arrays(A), oracle._csr_arrays(sp.csr_matrix(A.T)), w0, H0, lab, 4)
    assert len(secs) == 4 and rel_err(l32, l64) < 1e-6
