"""N2: the on-disk inputs of the reference's CPU engine, written by the reference's own tools
(preprocess + GCN-HP, see tests/golden/pargcn/README.md), feed the oracle and the engine."""
import io
import os
import re
import tarfile

import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.io import mmread

from conftest import GOLDEN, pkg, rel_err
from oracle import oracle

D = os.path.join(GOLDEN, "pargcn")


@pytest.fixture(scope="module")
def gemat_dir(tmp_path_factory):
    t = tmp_path_factory.mktemp("pargcn")
    with tarfile.open(os.path.join(D, "gemat11p_k3.tar.gz")) as tf:
        tf.extractall(t)
    return str(t / "out_gemat11p_k3")


@pytest.mark.parametrize("name,k", [("karate_k2", 2), ("karate_k3", 3)])
def test_read_directory(name, k):
    io_ = pkg("pargcn_io")
    prob = io_.load_directory(os.path.join(D, name))
    assert prob["k"] == k and prob["L"] == 3 and prob["d"] == [34, 16, 16, 2]
    # A.k values are the preprocess output printed with %.2f (GCN-HP/main.cpp:242)
    ref = sp.csr_matrix(mmread(os.path.join(D, "karate.A.mtx"))).toarray()
    got = sp.csr_matrix(prob["A"]).toarray()
    assert ((got != 0) == (np.round(ref, 2) != 0)).all() or ((got != 0) == (ref != 0)).all()
    assert np.abs(got - ref).max() <= 0.005 + 1e-6
    assert (prob["Y"][:, 1] == 1).all() and (prob["Y"][:, 0] == 0).all() and prob["Ymask"][:, 1].all()
    assert not prob["Ymask"][:, 0].any()                      # only column 1 is stored (GrB-GNN-IDG.py:76-78)
    assert sorted(np.bincount(prob["part"]).tolist()) == sorted(len(io_.read_rows(os.path.join(D, name, "H.%d" % p)))
                                                                for p in range(k))


@pytest.mark.parametrize("name,k", [("karate_k2", 2), ("karate_k3", 3)])
def test_boundary_sets_match_the_reference_partitioner(name, k):
    """conn.k / buff.k (written by GCN-HP's print_connectivity) == the send / receive sets our
    partition derives from the matrix alone."""
    io_, partition = pkg("pargcn_io"), pkg("partition")
    prob = io_.load_directory(os.path.join(D, name))
    A = prob["A"]
    for r in range(k):
        p = partition.build_partition(torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64)),
                                      torch.from_numpy(A.data.astype(np.float32)), A.shape[0],
                                      torch.from_numpy(prob["part"]), r, k)
        conn, nrecvs = prob["conn"][r]
        send_sizes, recv_sizes = prob["buff"][r]
        smap, rmap = p.send_map(), p.recv_map()
        assert {q for q, v in smap.items() if v.numel()} == set(conn)
        for q, ids in conn.items():
            np.testing.assert_array_equal(np.sort(ids), smap[q].numpy())
            assert send_sizes[q] == ids.size
        assert nrecvs == sum(1 for v in rmap.values() if v.numel())
        for q, cnt in recv_sizes.items():
            assert rmap[q].numel() == cnt


def test_cli_single_rank_matches_oracle(gemat_dir):
    """`pargcn.py -p DIR -c CONFIG` (host logic + printed lines) with the checker-backed kernels."""
    from oracle_kernels import OracleKernels
    pargcn, io_ = pkg("pargcn"), pkg("pargcn_io")
    buf = io.StringIO()
    os.environ["PGCN_SEED"] = "5"
    errs, Wn, Hout, part = pargcn.main(["-p", gemat_dir, "-c", os.path.join(gemat_dir, "config"), "-t", "4"],
                                       kernels=OracleKernels(), out=buf)
    prob = io_.load_directory(gemat_dir)
    d = prob["d"]
    n = d[0]
    A, dropped = oracle.drop_undelivered(prob["A"], prob["part"], prob["conn"], prob["k"])      # unsymmetric pattern
    assert dropped > 0
    W0 = pargcn.init_weights(d, 5)
    err, Wc, Hl, st = oracle.pargcn_train(A, [0] * n, 1, d, W0, np.ones((n, d[1]), np.float32),
                                          prob["Y"], prob["Ymask"])
    np.testing.assert_allclose(errs, err, rtol=1e-5)
    for l in Wc:
        assert rel_err(Wn[l].numpy(), Wc[l]) < 1e-5
    assert rel_err(Hout.numpy(), Hl[part.owned.numpy()]) < 1e-5
    out = buf.getvalue().splitlines()
    assert out[0] == "nlayers:3" and out[1].split() == [str(x) for x in d]
    printed = [float(l[4:]) for l in out if l.startswith("err:")]
    np.testing.assert_allclose(printed, err, rtol=1e-5)
    assert re.match(r"time : [0-9.]+ secs", out[5]) and out[6] == "0 0 0 0 0 0 0 0"


def test_io_errors(tmp_path):
    io_ = pkg("pargcn_io")
    (tmp_path / "config").write_text("3 4 2 2")
    with pytest.raises(ValueError):
        io_.read_config(str(tmp_path / "config"))
    (tmp_path / "A.0").write_text("4 2\n0 1 0.5\n")
    with pytest.raises(ValueError):
        io_.read_matrix(str(tmp_path / "A.0"))
    (tmp_path / "config").write_text("3 4 2 2 2")
    with pytest.raises(FileNotFoundError):
        io_.load_directory(str(tmp_path / "nothing"), str(tmp_path / "config"))
