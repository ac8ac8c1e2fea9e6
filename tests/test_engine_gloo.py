"""Multi-process CPU tests of the N>1 path (gloo, world_size 1..3).

The HOST logic under test is the product's: partition layout, all-to-all-v ordering,
PSpMM forward/backward structure, accumulate-on-receive, statistics, run().  The device
kernels are replaced by the checker-backed provider in tests/oracle_kernels.py (the only
way to execute on a box without a GPU; the real kernels are covered by -m gpu tests).
Outputs are compared with golden vectors produced by the reference's own PGCN.py."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch.multiprocessing as mp
from scipy.io import mmread

import _workers
from conftest import SPMM_CASES, SPMM_CASES_MORE, TRAIN_CASES, TRAIN_CASES_MORE, free_port, golden, golden_inputs, gpath, pkg, read_partvec, rel_err
from oracle import oracle

TOL = 1e-5


def _spawn(fn, P, *args, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=fn, args=(r, P, port) + args + (q,), kwargs=kw) for r in range(P)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(P)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r["rank"])


@pytest.mark.parametrize("name,mtx,pv,P", [SPMM_CASES[i] for i in (0, 2, 3, 4, 5, 7, 8)] + SPMM_CASES_MORE)
def test_pspmm_forward_backward(name, mtx, pv, P):
    arrays, meta = golden(name)
    res = _spawn(_workers.pspmm_worker, P, gpath(mtx), gpath(pv), meta["f"], meta["seed"])
    A = sp.csr_matrix(mmread(gpath(mtx))).astype(np.float32)
    n = A.shape[0]
    _, G = golden_inputs(n, meta["f"], meta["seed"])
    fwd = np.zeros((n, meta["f"]), np.float32)
    bwd = np.zeros_like(fwd)
    for r in res:
        fwd[r["own"]] = r["fwd"]
        bwd[r["own"]] = r["bwd"]
        m = meta["ranks"][r["rank"]]
        assert r["stats_fwd"] == m["stats_fwd"]                  # rows + messages, PGCN.py:105-114
        assert r["ok_halo"]
        # after fwd + bwd + one more forward exchange: 3 exchanges
        assert r["stats_all"]["send_nmsg"] == 3 * (P - 1)
        assert r["stats_all"]["send_volume"] == 2 * m["stats_fwd"]["send_volume"] + m["stats_fwd"]["recv_volume"]
        # the exchange probe of bench.py's N > 1 line (r06): one record per round and direction, bytes from the slab offsets
        ex = r["exchange"]
        if P == 1:
            assert ex == {}
            continue
        f = meta["f"]
        assert sorted(ex) == ["allreduce", "backward", "forward"] and ex["allreduce"]["bytes"] == 3 * f * 4 and ex["allreduce"]["calls"] == 1
        for tag, so, ro in (("forward", r["round_send_off"], r["round_recv_off"]), ("backward", r["round_recv_off"], r["round_send_off"])):
            assert [e["round"] for e in ex[tag]] == list(range(r["rounds"]))
            for e in ex[tag]:
                k = e["round"]
                assert e["calls"] == 1 and e["bytes_out"] == (so[k][-1] - so[k][0]) * f * 4 and e["bytes_in"] == (ro[k][-1] - ro[k][0]) * f * 4
                peers = [q for q in range(P) if q != r["rank"]]
                assert e["max_peer_bytes"] == max(max(so[k][q + 1] - so[k][q], ro[k][q + 1] - ro[k][q]) for q in peers) * f * 4
                assert e["ms"] >= 0 and e["exposed_ms"] == e["ms"]          # host-staged transport: nothing overlaps it
                assert abs(e["GBs_per_link"] - e["max_peer_bytes"] / max(e["ms"], 1e-12) / 1e6) <= 1e-6 * max(e["GBs_per_link"], 1) or e["ms"] == 0
                assert abs(e["frac_of_153GBs"] - e["GBs_per_link"] / 153.0) < 1e-9 and e["bound_ms_at_link_rate"] >= 0
    assert rel_err(fwd, arrays["fwd"]) < TOL                     # reference PSpMM.forward
    if "bwd" in arrays:
        assert rel_err(bwd, arrays["bwd"]) < TOL                 # reference PSpMM.backward (P <= 2)
    # P = 3: the reference overwrites partial sums (quirk Q3); the exact answer is A^T.G
    assert rel_err(bwd, oracle.spmm(sp.csr_matrix(A.T), G)) < TOL


@pytest.mark.parametrize("name,mtx,pv", TRAIN_CASES + TRAIN_CASES_MORE)
def test_run_matches_reference_training_p1(name, mtx, pv):
    arrays, meta = golden(name)
    res = _spawn(_workers.run_worker, 1, gpath(mtx), gpath(pv), meta["nlayers"], meta["f"], meta["seed"])
    out = res[0]["stdout"]
    printed = [float(x) for x in re.findall(r"Epoch \d{5} \| Loss ([0-9.]+)", out)]
    assert len(printed) == 4
    np.testing.assert_allclose(printed, arrays["losses"][1:], rtol=2e-5, atol=6e-5)
    for i, w in enumerate(res[0]["weights"]):
        assert rel_err(w, arrays["w1_%d" % i]) < 1e-4
    assert "total_vol: 0 total_nmsg: 0" in out and "Elapsed time" in out
    assert "'send_volume': tensor(0)" in out


@pytest.mark.parametrize("mtx,pv,P,L,f", [("karate.A.mtx", "karate.mtx.3.hp", 3, 3, 16),
                                          ("gemat11p.A.mtx", "gemat11.mtx.2.rp", 2, 2, 8)])
def test_run_multi_rank_matches_oracle(mtx, pv, P, L, f):
    seed = 7
    res = _spawn(_workers.run_worker, P, gpath(mtx), gpath(pv), L, f, seed)
    A = sp.csr_matrix(mmread(gpath(mtx))).astype(np.float32)
    n = A.shape[0]
    part = read_partvec(gpath(pv))
    import torch
    import torch.nn as nn
    torch.manual_seed(seed)
    w0 = [nn.Linear(f, f, bias=False).weight.detach().numpy() for _ in range(L)]
    H0 = np.repeat(np.arange(n, dtype=np.float32)[:, None], f, axis=1)
    losses, Ws = oracle.pgcn_train_np(A, part, P, w0, H0, np.arange(n) % f, epochs=5)
    printed = [float(x) for x in re.findall(r"Epoch \d{5} \| Loss ([0-9.]+)", res[0]["stdout"])]
    np.testing.assert_allclose(printed, losses[1:], rtol=5e-5, atol=6e-5)
    for r in res:
        for i, w in enumerate(r["weights"]):
            assert rel_err(w, Ws[i]) < 2e-4
    # volume KAT: rows per exchange x (1 + 4 epochs) x L layers x 2 directions
    rows = sum(v.size for r in range(P) for v in oracle.communication_maps(A, part, r, P)[0].values())
    m = re.search(r"total_vol: (\d+) total_nmsg: (\d+)", res[0]["stdout"])
    assert int(m.group(1)) == rows * 5 * L * 2
    assert int(m.group(2)) == P * (P - 1) * 5 * L * 2


def test_total_vol_kat_gemat11_hp3():
    """the reference prints total_vol 55380 / total_nmsg 180 for
    PGCN.py -s 3 -l 3 -f 128 on gemat11 with the shipped .3.hp part vector."""
    res = _spawn(_workers.run_worker, 3, gpath("gemat11.mtx"), gpath("gemat11.mtx.3.hp"), 3, 8, 1)
    assert "total_vol: 55380 total_nmsg: 180" in res[0]["stdout"]


def test_pargcn_semantics_multi_rank():
    d = [4929, 16, 16, 2]
    seed = 3
    res = _spawn(_workers.pargcn_worker, 3, gpath("gemat11p.mtx"), gpath("gemat11.mtx.3.hp"), d, seed)
    A = oracle.normalize_adjacency(mmread(gpath("gemat11p.mtx")))
    A = ((A + A.T) * 0.5).tocsr().astype(np.float32)
    n = d[0]
    rng = np.random.default_rng(seed)
    W = {l: (rng.random((d[l], d[l + 1]), dtype=np.float32) * 2 - 1) * np.float32(np.sqrt(6.0 / (d[l] + d[l + 1])))
         for l in (1, 2)}
    Y = np.zeros((n, 2), np.float32); Y[:, 1] = 1
    Ym = np.zeros((n, 2), np.uint8); Ym[:, 1] = 1
    part = read_partvec(gpath("gemat11.mtx.3.hp"))
    err, Wc, Hl, st = oracle.pargcn_train(A, part, 3, d, W, np.ones((n, 16), np.float32), Y, Ym)
    Hg = np.zeros((n, 2), np.float32)
    for r in res:
        np.testing.assert_allclose(r["errs"], err, rtol=1e-5)
        for l in (1, 2):
            assert rel_err(r["W"][l], Wc[l]) < TOL
        Hg[r["own"]] = r["Hl"]
        # engine counts ROWS per exchange (PGCN.py:105), the oracle SCALARS (main.c:264):
        # 12 exchanges of rows_r rows vs rows_r x (16+16+2+16) scalars x 3 epochs
        assert r["stats"]["send_volume"] % 12 == 0
        assert r["stats"]["send_volume"] // 12 * 50 * 3 == st[r["rank"], 0]
        assert r["stats"]["send_nmsg"] == st[r["rank"], 1]
    assert rel_err(Hg, Hl) < TOL


@pytest.mark.parametrize("rounds", [1, 2])
@pytest.mark.parametrize("name,mtx,pv,P", [SPMM_CASES[i] for i in (0, 3, 4, 6, 8)])
def test_partition_from_own_rows_equals_global_build(name, mtx, pv, P, rounds):
    """N1: every rank loads ONLY its rows (pgcn_load_mtx_partition) and builds its Partition with two small
    collectives (degree all-reduce, all-to-all-v of needed ids) -- field for field what the global scan gives."""
    _, meta = golden(name)
    res = _spawn(_workers.partition_local_worker, P, gpath(mtx), gpath(pv), rounds)
    for r in res:
        assert all(r["checks"].values()), (r["rank"], r["checks"])
        assert r["nnz_local"] == meta["ranks"][r["rank"]]["nnz_local"]


def test_run_with_row_block_ingest_is_identical(monkeypatch):
    """PGCN_INGEST=rows: run() where no rank parses more than its own rows into memory -- same printed
    losses, same statistics, bit-identical weights as the default (global) ingest."""
    mtx, pv, P, L, f = "gemat11p.A.mtx", "gemat11.mtx.3.hp", 3, 2, 8
    monkeypatch.setenv("PGCN_INGEST", "global")           # the reference's way: every rank parses the whole matrix
    base = _spawn(_workers.run_worker, P, gpath(mtx), gpath(pv), L, f, 7)
    monkeypatch.setenv("PGCN_INGEST", "rows")             # (the default since r02)
    rows = _spawn(_workers.run_worker, P, gpath(mtx), gpath(pv), L, f, 7)
    assert rows[0]["stdout"].split("Elapsed")[0] == base[0]["stdout"].split("Elapsed")[0]      # env echo, losses, stats
    vol = re.findall(r"total_vol: \d+ total_nmsg: \d+", rows[0]["stdout"])
    assert len(vol) == 1 and vol == re.findall(r"total_vol: \d+ total_nmsg: \d+", base[0]["stdout"])
    for a, b in zip(rows, base):
        for wa, wb in zip(a["weights"], b["weights"]):
            np.testing.assert_array_equal(wa, wb)


def test_run_from_binary_csr_shards_is_identical(tmp_path, monkeypatch):
    """`-a PREFIX` with PREFIX.<rank>.pgcsr shards (pgcn_shard_*): every rank reads only its own binary row block;
    same printed losses / statistics / weights as the MatrixMarket run."""
    from scipy.io import mmread
    ingest = pkg("ingest")
    mtx, pv, P, L, f = "gemat11p.A.mtx", "gemat11.mtx.3.hp", 3, 2, 8
    monkeypatch.setenv("PGCN_INGEST", "global")
    base = _spawn(_workers.run_worker, P, gpath(mtx), gpath(pv), L, f, 7)
    prefix = str(tmp_path / "gemat11p")
    paths = ingest.write_shards(prefix, mmread(gpath(mtx)), read_partvec(gpath(pv)), P)
    assert [os.path.basename(x) for x in paths] == ["gemat11p.%d.pgcsr" % r for r in range(P)]
    sh = _spawn(_workers.run_worker, P, prefix, gpath(pv), L, f, 7)
    assert sh[0]["stdout"].split("Elapsed")[0] == base[0]["stdout"].split("Elapsed")[0]
    for a, b in zip(sh, base):
        for wa, wb in zip(a["weights"], b["weights"]):
            np.testing.assert_array_equal(wa, wb)


@pytest.mark.parametrize("case,P", [("karate_k2", 2), ("karate_k3", 3)])
def test_pargcn_main_multi_rank(case, P):
    """`pargcn.main` (the grbgcn command line, Parallel-GCN/main.c:120-165, :441-454, statistics
    :506-524) with world > 1 on a directory the reference's own GCN-HP tool wrote: rank 0 prints the reference's
    lines, every rank ends with the same weights, the numbers are the oracle's P-rank training loop."""
    import re
    from conftest import GOLDEN
    io_, pargcn = pkg("pargcn_io"), pkg("pargcn")
    directory = os.path.join(GOLDEN, "pargcn", case)
    seed = 5
    res = _spawn(_workers.pargcn_main_worker, P, directory, seed)
    prob = io_.load_directory(directory)
    d, n = prob["d"], prob["d"][0]
    A = sp.csr_matrix(prob["A"])
    err, Wc, Hl, st = oracle.pargcn_train(A, prob["part"], P, d, pargcn.init_weights(d, seed), np.ones((n, d[1]), np.float32),
                                          prob["Y"], prob["Ymask"])
    out0 = res[0]["stdout"]
    assert out0.startswith("nlayers:%d\n" % prob["L"]) and (" ".join(str(x) for x in d) + " ") in out0
    printed = [float(x) for x in re.findall(r"^err:(\S+)$", out0, re.M)]
    assert len(printed) == 3
    np.testing.assert_allclose(printed, err, rtol=1e-4)
    assert re.search(r"^time : \d+\.\d+ secs$", out0, re.M)
    for r in range(P):
        np.testing.assert_allclose(res[r]["errs"], err, rtol=1e-5)
        for l in Wc:
            assert rel_err(res[r]["W"][l], Wc[l]) < 1e-5
            assert np.array_equal(res[r]["W"][l], res[0]["W"][l])          # replicas stay identical
        assert rel_err(res[r]["H"], Hl[res[r]["own"]]) < 1e-5
        if r > 0:
            assert "err:" not in res[r]["stdout"] and "time :" not in res[r]["stdout"]
    # the statistics line (main.c:506-524): volumes in scalars, three epochs, widths f..f forward and 2, f.. backward
    L = prob["L"]
    widths = [d[l] for l in range(1, L)] + [d[l + 1] for l in range(L - 1, 0, -1)]
    vol = [res[r]["n_send"] * sum(widths) * 3 for r in range(P)]
    rcv = [res[r]["n_halo"] * sum(widths) * 3 for r in range(P)]
    msg = [res[r]["targets"] * len(widths) * 3 for r in range(P)]
    msr = [res[r]["sources"] * len(widths) * 3 for r in range(P)]
    last = [int(x) for x in out0.strip().splitlines()[-1].split()]
    assert last == [sum(vol), sum(vol) // P, max(vol), max(rcv), sum(msg), sum(msg) // P, max(msg), max(msr)]
