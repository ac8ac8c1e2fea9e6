"""GAT path (N3): the sparse numpy restatement in oracle/ is pinned to outputs of the reference's own
dense PGAT layers (tests/golden/ref_gat_*, made by make_golden_gat.py from /root/reference/GPU/PGAT.py),
and the "standard" mode (edge softmax over neighbours, K heads) to a dense torch autograd model."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.io import mmread

from conftest import golden, gpath, rel_err
from oracle import oracle


def positive_pattern(A):
    """PGAT.py:146 masks with ``A > 0`` on the DENSE row block (duplicates summed by to_dense)."""
    A = sp.csr_matrix(A)
    A.sum_duplicates()
    A = A.multiply(A > 0).tocsr()
    A.eliminate_zeros()
    A.data[:] = 1
    return A


def chain_forward_backward(A, arrays, L, dtype):
    """L reference-mode layers + run()'s objective, forward and backward, in numpy."""
    n = A.shape[0]
    H = arrays["H"].astype(dtype)
    f = H.shape[1]
    x, saved, outs = H, [], []
    for i in range(L):
        W, a = arrays["W_%d" % i].astype(dtype), arrays["a_%d" % i].astype(dtype)
        out, Z, s1, s2 = oracle.gat_layer_np(A, x, W, a, 1, "reference")
        saved.append((x, W, a, Z, s1, s2))
        outs.append(out)
        x = out
    logits = x
    z = logits - logits.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    labels = np.arange(n) % f
    loss = -logp[np.arange(n), labels].mean()
    g = np.exp(logp)
    g[np.arange(n), labels] -= 1
    g /= n
    grads = {}
    for i in reversed(range(L)):
        x, W, a, Z, s1, s2 = saved[i]
        grads["dout_%d" % i] = g
        dZ, ds1, ds2 = oracle.gat_aggregate_backward_np(A, Z, s1, s2, g, "reference")
        F = Z.shape[1]
        dZ = dZ + ds1 @ a[:F].T + ds2 @ a[F:].T
        grads["da_%d" % i] = np.concatenate([Z.T @ ds1, Z.T @ ds2])
        grads["dW_%d" % i] = dZ.T @ x
        g = dZ @ W
    grads["dH"] = g
    return outs, loss, grads


@pytest.mark.parametrize("name,mtx", [("ref_gat_karateA", "karate.A.mtx"), ("ref_gat_gemat11pA", "gemat11p.A.mtx"),
                                      ("ref_gat_coraA", "cora.A.mtx")])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-5), (np.float32, 1e-4)])
def test_sparse_restatement_matches_reference_layers(name, mtx, dtype, tol):
    arrays, meta = golden(name)
    A = positive_pattern(mmread(gpath(mtx)))
    assert A.nnz == meta["edges_positive"]
    outs, loss, grads = chain_forward_backward(A, arrays, meta["layers"], dtype)
    for i, o in enumerate(outs):
        assert rel_err(o, arrays["out_%d" % i]) < tol
    assert abs(loss - meta["loss"]) < tol * abs(meta["loss"])
    for k, g in grads.items():
        assert rel_err(g, arrays[k]) < 20 * tol, k      # gradients pass through two softmaxes in fp32 on the reference side


def test_literal_dense_arithmetic_agrees():
    rng = np.random.default_rng(3)
    n, f = 50, 5
    A = sp.random(n, n, density=0.1, random_state=2, format="csr")
    A.data[:] = rng.standard_normal(A.nnz)                 # negative entries are NOT edges (A > 0)
    H, W, a = rng.standard_normal((n, f)), rng.standard_normal((f, f)), rng.standard_normal((2 * f, 1))
    out, *_ = oracle.gat_layer_np(positive_pattern(A), H, W, a, 1, "reference")
    assert rel_err(out, oracle.gat_dense_reference_np(A.toarray(), H, W, a)) < 1e-12


@pytest.mark.parametrize("K,d", [(1, 8), (3, 4), (4, 16)])
def test_standard_mode_against_dense_autograd(K, d):
    rng = np.random.default_rng(K)
    n = 60
    A = sp.random(n, n, density=0.08, random_state=K, format="csr")
    A.data[:] = 1
    A = A.tolil(); A[7] = 0; A = A.tocsr(); A.eliminate_zeros()            # an empty row
    Z, s1, s2 = rng.standard_normal((n, K * d)), rng.standard_normal((n, K)), rng.standard_normal((n, K))
    G = rng.standard_normal((n, K * d))
    out = oracle.gat_aggregate_np(A, Z, s1, s2, "standard", slope=0.2)
    dZ, ds1, ds2 = oracle.gat_aggregate_backward_np(A, Z, s1, s2, G, "standard", slope=0.2)
    Zt, s1t, s2t = (torch.tensor(x, requires_grad=True) for x in (Z, s1, s2))
    M = torch.tensor(A.toarray()) > 0
    e = torch.nn.functional.leaky_relu(s1t[:, None, :] + s2t[None, :, :], 0.2)
    e = torch.where(M[:, :, None], e, torch.full_like(e, -1e300))
    al = torch.softmax(e, 1) * M[:, :, None]
    ot = torch.einsum("ijk,jkd->ikd", al, Zt.view(n, K, d)).reshape(n, K * d)
    ot.backward(torch.tensor(G))
    assert rel_err(out, ot.detach().numpy()) < 1e-12
    assert np.abs(out[7]).max() == 0
    assert rel_err(dZ, Zt.grad.numpy()) < 1e-12
    assert rel_err(ds1, s1t.grad.numpy()) < 1e-12
    assert rel_err(ds2, s2t.grad.numpy()) < 1e-12
