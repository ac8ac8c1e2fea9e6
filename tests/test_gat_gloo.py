"""GAT path, host logic on CPU (gloo, world_size 1..3): partition -> combined [local ; halo] structure,
[Z | s2] boundary exchange, edge softmax / weighted SpMM / edge gradient call structure, reverse
exchange with accumulation, the two small all-reduces of the reference mode, run().  Kernels are the
checker-backed stand-ins of tests/oracle_kernels.py; the HIP kernels are covered by the -m gpu tests.
Expected values: the reference's own PGAT layers (tests/golden/ref_gat_*) and the numpy oracle."""
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.io import mmread

import _workers
from conftest import golden, gpath, pkg, rel_err
from oracle import oracle
from test_engine_gloo import _spawn
from test_gat_oracle import positive_pattern


def _pattern(mtx, mode):
    A = sp.csr_matrix(mmread(gpath(mtx)))
    if mode == "reference":
        return positive_pattern(A)
    A.sum_duplicates()
    A.data[:] = 1
    return A


def _expected(A, mode, heads, f, L, seed):
    """The same seeded model in float64 numpy on the whole graph (one process)."""
    n = A.shape[0]
    rng = np.random.default_rng(seed)
    H = (rng.random((n, f), dtype=np.float32) * 2 - 1).astype(np.float64)
    d = f // heads
    Ws, As = [], []
    for _ in range(L):
        Ws.append((rng.standard_normal((f, f)) * 0.4).astype(np.float32).astype(np.float64))
        As.append((rng.standard_normal((2 * d, heads)) * 0.4).astype(np.float32).astype(np.float64))
    x, saved, outs = H, [], []
    for W, a in zip(Ws, As):
        out, Z, s1, s2 = oracle.gat_layer_np(A, x, W, a, heads, mode)
        saved.append((x, W, a, Z, s1, s2))
        outs.append(out)
        x = out
    z = x - x.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    labels = np.arange(n) % f
    loss = -logp[np.arange(n), labels].mean()
    g = np.exp(logp)
    g[np.arange(n), labels] -= 1
    g /= n
    dW, da = [None] * L, [None] * L
    for i in reversed(range(L)):
        x, W, a, Z, s1, s2 = saved[i]
        dZ, ds1, ds2 = oracle.gat_aggregate_backward_np(A, Z, s1, s2, g, mode)
        Zh = Z.reshape(n, heads, d)
        dZ = dZ + (ds1[:, :, None] * a[:d].T[None] + ds2[:, :, None] * a[d:].T[None]).reshape(n, f)
        da[i] = np.concatenate([np.einsum("nkd,nk->dk", Zh, ds1), np.einsum("nkd,nk->dk", Zh, ds2)])
        dW[i] = dZ.T @ x
        g = dZ @ W
    return outs, loss, g, dW, da


CASES = [
    # mtx, part vector, P, mode, heads, f, L
    ("karate.A.mtx", "karate.mtx.1.rp", 1, "reference", 1, 16, 2),
    ("karate.A.mtx", "karate.mtx.3.hp", 3, "reference", 1, 16, 2),
    ("karate.A.mtx", "karate.mtx.2.rp", 2, "standard", 4, 16, 2),
    ("karate.mtx", "karate.mtx.3.stchp", 3, "standard", 1, 6, 3),
    ("gemat11p.A.mtx", "gemat11.mtx.3.hp", 3, "standard", 2, 8, 2),
    ("gemat11p.A.mtx", "gemat11.mtx.2.rp", 2, "reference", 1, 8, 2),
    ("gemat11.mtx", "gemat11.mtx.3.rp", 3, "reference", 1, 5, 1),     # negative entries: A > 0 masks them (PGAT.py:146)
    # head widths the two-pass route covers (d = 32 / 64: forward with the second accumulator, edge gradient inside the
    # transposed product, ds1 = <dOut, V> - t C -- gat.GatEngine on the stand-ins of tests/oracle_kernels.py)
    ("karate.A.mtx", "karate.mtx.2.rp", 2, "standard", 2, 64, 2),
    ("gemat11p.A.mtx", "gemat11.mtx.3.hp", 3, "standard", 1, 64, 2),
    ("gemat11p.A.mtx", "gemat11.mtx.2.rp", 2, "reference", 1, 32, 2),
]


@pytest.mark.parametrize("mtx,pv,P,mode,heads,f,L", CASES)
def test_layers_forward_backward_any_partition(mtx, pv, P, mode, heads, f, L):
    seed = 11
    res = _spawn(_workers.gat_layers_worker, P, gpath(mtx), gpath(pv), mode, heads, f, L, seed)
    A = _pattern(mtx, mode)
    n = A.shape[0]
    outs, loss, dH, dW, da = _expected(A, mode, heads, f, L, seed)
    got_out = [np.zeros((n, f), np.float32) for _ in range(L)]
    got_dH = np.zeros((n, f), np.float32)
    for r in res:
        for i in range(L):
            got_out[i][r["own"]] = r["outs"][i]
        got_dH[r["own"]] = r["dH"]
        assert r["ok_halo"]
        d = f // heads                                        # the two-pass route runs exactly where it is covered
        assert r["fused"] == [d in (32, 64, 128, 256) and f <= 256] * L
    for i in range(L):
        assert rel_err(got_out[i], outs[i]) < 2e-5
    assert abs(sum(r["loss"] for r in res) - loss) < 1e-5 * abs(loss)       # SUM over ranks == one-process loss
    assert rel_err(got_dH, dH) < 2e-4
    for i in range(L):                                                       # parameter grads: SUM over ranks
        assert rel_err(sum(r["dW"][i] for r in res), dW[i]) < 2e-4
        assert rel_err(sum(r["da"][i] for r in res), da[i]) < 2e-4
    # Comm.backward: d(sum of halo rows)/dH = how many peers receive each owned row
    part = np.array(open(gpath(pv)).readline().split(), dtype=np.int64)
    Ag = sp.csr_matrix(mmread(gpath(mtx)))
    for r in res:
        cnt = np.zeros(n)
        coo = Ag.tocoo()
        mask = (part[coo.col] == r["rank"]) & (part[coo.row] != r["rank"])
        pairs = np.unique(np.stack([part[coo.row][mask], coo.col[mask]]), axis=1)
        np.add.at(cnt, pairs[1], 1)
        np.testing.assert_array_equal(r["comm_grad"], np.repeat(cnt[r["own"]][:, None], f, 1).astype(np.float32))
        assert r["n_send_rows"] == pairs.shape[1]


@pytest.mark.parametrize("name,mtx", [("ref_gat_karateA", "karate.A.mtx"), ("ref_gat_gemat11pA", "gemat11p.A.mtx"),
                                      ("ref_gat_coraA", "cora.A.mtx")])
def test_reference_mode_reproduces_reference_layers(name, mtx):
    """The product's host path (P = 1) in reference mode against the outputs and gradients of the
    reference's own dense PGAT layers -- same parameters, same input."""
    arrays, meta = golden(name)
    M = _workers._pgat_module(0, 1, "reference", 1)
    A = mmread(gpath(mtx))
    n, f, L = meta["n"], meta["f"], meta["layers"]
    eng = M.get_partitiont_of_adjacency_matrix(A, [0] * n, 0)
    own = eng.part.owned.numpy()
    H = torch.tensor(arrays["H"][own], requires_grad=True)
    layers = [M.PGAT(eng, f, f) for _ in range(L)]
    with torch.no_grad():
        for i, layer in enumerate(layers):
            layer.linear.weight.copy_(torch.from_numpy(arrays["W_%d" % i]))
            layer.attention.copy_(torch.from_numpy(arrays["a_%d" % i]))
    x = H
    for i, layer in enumerate(layers):
        x = layer(x)
        assert rel_err(x.detach().numpy(), arrays["out_%d" % i][own]) < 1e-4
    loss = M.local_loss(x, torch.from_numpy(own) % f, n)
    assert abs(float(loss.detach()) - meta["loss"]) < 1e-5 * meta["loss"]
    loss.backward()
    assert rel_err(H.grad.numpy(), arrays["dH"][own]) < 2e-3
    for i, layer in enumerate(layers):
        assert rel_err(layer.linear.weight.grad.numpy(), arrays["dW_%d" % i]) < 2e-3
        assert rel_err(layer.attention.grad.numpy(), arrays["da_%d" % i]) < 2e-3


def _losses(stdout):
    return [float(x) for x in re.findall(r"Epoch \d{5} \| Loss ([-\d.naninf]+)", stdout)]


def test_run_reproduces_reference_run_p1_and_p3():
    """run() (PGAT.py:165-233) in reference mode: the printed losses of the reference's own run at
    P = 1, from one rank and from three ranks (which must agree with each other more tightly)."""
    _, meta = golden("ref_gat_run_karateA")
    f, L, seed = meta["f"], meta["layers"], meta["seed"]
    r1 = _spawn(_workers.gat_run_worker, 1, gpath(meta["mtx"]), gpath("ref_gat_run_karateA.partvec"), "reference", 1,
                L, f, seed, 50)
    r3 = _spawn(_workers.gat_run_worker, 3, gpath(meta["mtx"]), gpath("karate.mtx.3.hp"), "reference", 1, L, f, seed, 50)
    l1, l3 = _losses(r1[0]["stdout"]), _losses(r3[0]["stdout"])
    assert len(l1) == len(l3) == len(meta["losses"]) == 50
    assert "Elapsed time" in r1[0]["stdout"]
    # H[i,:] = i (PGAT.py:196) reaches 33 and the fp32 trajectories drift apart slowly: the first
    # epochs agree to the printed digits, all 50 to 1 %
    np.testing.assert_allclose(l1[:10], meta["losses"][:10], rtol=1e-5, atol=1.5e-4)
    np.testing.assert_allclose(l1, meta["losses"], rtol=1e-2)
    np.testing.assert_allclose(l3[:10], l1[:10], rtol=1e-5, atol=1.5e-4)
    np.testing.assert_allclose(l3, l1, rtol=1e-2)
    arrays, _ = golden("ref_gat_run_karateA")
    for k, v in r1[0]["params"].items():
        assert rel_err(v, arrays["final_" + k.replace(".", "_")]) < 2e-2
    for k, v in r3[0]["params"].items():                          # replicas stay identical
        for other in r3[1:]:
            np.testing.assert_array_equal(v, other["params"][k])


def test_module_contract():
    M = pkg("PGAT")
    for name in ("compute_communication_maps", "get_partitiont_of_adjacency_matrix", "communicate_fgm", "Comm", "PGAT",
                 "average_gradients", "initiliaze_parameters", "run", "init_process", "main"):
        assert hasattr(M, name)
    layer = M.PGAT(None, 12, 12, heads=1)
    assert tuple(layer.attention.shape) == (24, 1) and tuple(layer.linear.weight.shape) == (12, 12)   # PGAT.py:126-127
    assert tuple(M.PGAT(None, 12, 12, heads=3).attention.shape) == (8, 3)
    with pytest.raises(ValueError):
        M.PGAT(None, 12, 10, heads=4)


def test_four_ranks_random_partition(tmp_path):
    """More peers than the shipped part vectors have: P = 4, seeded random partition, 2 heads."""
    mtx, mode, heads, f, L, seed = "gemat11p.A.mtx", "standard", 2, 8, 2, 5
    A = _pattern(mtx, mode)
    n = A.shape[0]
    part = np.random.default_rng(9).integers(0, 4, n)
    pv = tmp_path / "rand4.pv"
    pv.write_text(" ".join(map(str, part.tolist())) + "\n")
    res = _spawn(_workers.gat_layers_worker, 4, gpath(mtx), str(pv), mode, heads, f, L, seed)
    outs, loss, dH, dW, da = _expected(A, mode, heads, f, L, seed)
    got = np.zeros((n, f), np.float32)
    got_dH = np.zeros((n, f), np.float32)
    for r in res:
        assert np.array_equal(np.sort(r["own"]), np.nonzero(part == r["rank"])[0]) and r["ok_halo"]
        got[r["own"]] = r["outs"][-1]
        got_dH[r["own"]] = r["dH"]
    assert rel_err(got, outs[-1]) < 2e-5
    assert abs(sum(r["loss"] for r in res) - loss) < 1e-5 * abs(loss)
    assert rel_err(got_dH, dH) < 2e-4
    for i in range(L):
        assert rel_err(sum(r["dW"][i] for r in res), dW[i]) < 2e-4
        assert rel_err(sum(r["da"][i] for r in res), da[i]) < 2e-4


@pytest.mark.parametrize("mode", ["standard", "reference"])
def test_backward_survives_a_refusal_of_the_fused_gradient_kernel(mode):
    """After a fused forward there are no alpha planes.  When pgcn_spmm_heads_grad_f32 then refuses its operands (an
    alignment or stride its forward twin did not care about), GatEngine.backward rebuilds the planes and takes the
    unfused passes -- same gradients."""
    gat, partition, synth = pkg("gat"), pkg("partition"), pkg("synth")
    from oracle_kernels import OracleKernels
    n, row, col, val = synth.make_graph(400, 6000, seed=2)
    part = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64), 0, 1)
    heads, d = 2, 32
    g = torch.Generator().manual_seed(5)
    Z, s1, s2 = torch.rand(n, heads * d, generator=g) - 0.5, torch.rand(n, heads, generator=g) - 0.5, torch.rand(n, heads, generator=g) - 0.5
    dOut = torch.rand(n, heads * d, generator=g) - 0.5
    res = []
    for refuse in (False, True):
        K = OracleKernels()
        eng = gat.GatEngine(part, K, torch.device("cpu"), None, mode=mode)
        st = eng.new_layer_state(heads, d)
        eng.forward(st, Z, s1, s2)
        assert st.fused and st.alpha is None
        if refuse:
            real = K.spmm_heads_grad
            K.spmm_heads_grad = lambda *a, **k: False
        out = eng.backward(st, dOut)
        assert (st.alpha is not None) == refuse and st.fused != refuse
        res.append([t.clone() for t in out])
    for a, b in zip(*res):
        assert rel_err(b.numpy(), a.numpy()) < 1e-5
