"""The input kinds of bench.py: --shards PREFIX (every rank reads only its own binary CSR
shard: the papers100M-scale path, replacing the whole-matrix parse of /root/reference/GPU/PGCN.py:171), --mtx FILE
(a real MatrixMarket file, picked up automatically from $PGCN_DATA_DIR), --emulate-rank r/P.  Host logic only: the
partitions they produce must be the ones build_partition derives from the global matrix."""
import argparse
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.multiprocessing as mp
from scipy.io import mmwrite

import _workers
from conftest import ROOT, pkg

sys.path.insert(0, ROOT)


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _graph(tmp_path, n=3000, nnz=60000, P=2):
    synth, ingest, io_ = pkg("synth"), pkg("ingest"), pkg("pargcn_io")
    n, row, col, val = synth.make_graph(n, nnz, seed=11)
    A = sp.coo_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    pv = synth.random_partvec(n, P, seed=5)
    mtx = str(tmp_path / "g.mtx")
    mmwrite(mtx, A, precision=9)
    pvf = str(tmp_path / ("g.mtx.%d.rp" % P))
    io_.write_partvec(pvf, pv.numpy())
    prefix = str(tmp_path / "g")
    ingest.write_shards(prefix, A, pv.numpy(), P)
    return n, row, col, val, pv, mtx, pvf, prefix


def _args(**kw):
    base = dict(shards=None, mtx=None, partvec="random", emulate_rank=None, workload="none", generator="rmat", real=False)
    base.update(kw)
    return argparse.Namespace(**base)


def _spawn(P, kind, path, pvf):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_workers.bench_inputs_worker, args=(r, P, port, kind, path, pvf, q)) for r in range(P)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(P):
        d = q.get(timeout=600)
        res[d["rank"]] = d
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("kind", ["shards", "mtx"])
def test_rank_local_inputs_give_the_global_partition(tmp_path, kind):
    """Two ranks over gloo, each reading ONLY its own rows (shard file / C++ row filter of the .mtx): field for
    field the partition that the global scan of GPU/PGCN.py:37-64 gives."""
    P = 2
    n, row, col, val, pv, mtx, pvf, prefix = _graph(tmp_path, P=P)
    res = _spawn(P, kind, prefix if kind == "shards" else mtx, pvf)
    partition = pkg("partition")
    for r in range(P):
        ref = partition.build_partition(row, col, val, n, pv, r, P)
        got = res[r]
        assert got["n"] == n and got["nnz"] == row.numel() and got["partition"] == "file:" + os.path.basename(pvf)
        assert np.array_equal(got["owned"], ref.owned.numpy())
        assert np.array_equal(got["halo_global"], ref.halo_global.numpy())
        assert np.array_equal(got["send_global"], ref.send_global.numpy())
        rr, cc, vv = ref.A_loc.to_coo()
        key = torch.argsort(rr * ref.n_local + cc, stable=True)
        assert np.array_equal(got["loc"][0], rr[key].numpy()) and np.array_equal(got["loc"][1], cc[key].numpy())
        assert np.array_equal(got["loc"][2].view(np.int32), vv[key].numpy().view(np.int32))    # values bit-exact
        assert got["nnz_halo"] == sum(a.nnz for a in ref.A_halo)


def test_single_rank_inputs_and_emulated_rank(tmp_path):
    import bench
    n, row, col, val, pv, mtx, pvf, prefix = _graph(tmp_path, P=1)
    partition = pkg("partition")
    ref = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64), 0, 1)
    dev = torch.device("cpu")
    for a in (_args(shards=prefix, partvec=pvf), _args(mtx=mtx)):
        part, info = bench.acquire_partition(a, 0, 1, dev, lambda m: None)
        assert info["n"] == n and info["nnz"] == row.numel()
        assert torch.equal(part.owned, ref.owned) and part.A_loc.nnz == ref.A_loc.nnz
    # --emulate-rank r/P on a file: rank r's pieces of a P-way random partition, built on one process
    part, info = bench.acquire_partition(_args(mtx=mtx, emulate_rank="1/4"), 0, 1, dev, lambda m: None)
    ref4 = partition.build_partition(row, col, val, n, pkg("synth").random_partvec(n, 4, seed=0), 1, 4)
    assert part.rank == 1 and part.size == 4 and torch.equal(part.owned, ref4.owned)
    assert torch.equal(part.halo_global, ref4.halo_global) and info["partition"].startswith("random")


def test_shards_of_another_part_vector_are_refused(tmp_path):
    import bench
    n, row, col, val, pv, mtx, pvf, prefix = _graph(tmp_path, P=2)
    other = str(tmp_path / "other.rp")
    pkg("pargcn_io").write_partvec(other, pkg("synth").random_partvec(n, 2, seed=99).numpy())
    with pytest.raises(SystemExit) as e:
        bench.acquire_partition(_args(shards=prefix, partvec=other, emulate_rank=None), 0, 1, torch.device("cpu"), lambda m: None)
    assert "written for rank" in str(e.value) or "does not hold the rows" in str(e.value)


def test_real_file_is_picked_up_from_data_dir(tmp_path, monkeypatch):
    import bench
    n, row, col, val, pv, mtx, pvf, prefix = _graph(tmp_path, P=1)
    os.rename(mtx, str(tmp_path / "reddit.mtx"))
    monkeypatch.setenv("PGCN_DATA_DIR", str(tmp_path))
    assert bench.real_mtx_for("reddit") == str(tmp_path / "reddit.mtx") and bench.real_mtx_for("products") is None
    part, info = bench.acquire_partition(_args(workload="reddit"), 0, 1, torch.device("cpu"), lambda m: None)
    assert info["data"] == "file" and info["n"] == n and "reddit.mtx" in info["source"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="flag parsing on a box without a GPU")
def test_new_flags_parse_and_fail_loudly_without_gpu(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shards", str(tmp_path / "x"), "--emulate-rank", "0/8",
                          "--mtx", str(tmp_path / "y.mtx"), "--real", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 1 and "needs an MI355X" in out.stderr and "unrecognized" not in out.stderr


@pytest.mark.parametrize("P,break_it", [(3, False), (2, True)])
def test_multirank_selftest_over_gloo(P, break_it):
    """bench.py's first-contact check of an N-rank job (runs before the timed region of every N > 1 run): P ranks through
    the real engine and exchange against the same kernels on one rank; a wrong received row fails it on EVERY rank."""
    from test_engine_gloo import _spawn
    res = _spawn(_workers.bench_selftest_worker, P, break_it)
    if break_it:
        assert all(not r["ok"] and "FAILED" in r["msg"] for r in res)
        return
    assert all(r["ok"] for r in res)
    rec = res[0]["rec"]
    assert rec["ranks"] == P and rec["rounds"] == 2 and rec["exchanger_selftest"] == {"torch.distributed": True}
    assert rec["forward_rel_err_vs_one_rank"] < 1e-5 and rec["backward_rel_err_vs_one_rank"] < 1e-5
    assert all(r["rec"]["boundary_rows_this_rank"] > 0 for r in res)
