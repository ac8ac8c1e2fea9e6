"""Host-side pieces added in round 2 (CPU): writers of the reference's on-disk formats (GCN-HP/main.cpp:117-282,
GPU/hypergraph/main.cpp:51-63), the strip-tile layout, the second synthetic generator and the vertex order."""
import os
import tarfile

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import GOLDEN, gpath, pkg


def _same_file(a, b, any_line_order=False):
    """Byte-identical, except where the reference iterates an unordered_map (rows of A.k, peers of conn.k /
    buff.k): there the same header and the same set of lines / (peer, count) pairs."""
    ta, tb = open(a).read(), open(b).read()
    name = os.path.basename(a)
    if any_line_order or name.startswith("conn."):
        la, lb = ta.split("\n"), tb.split("\n")
        return la[0] == lb[0] and sorted(la[1:]) == sorted(lb[1:])
    if name.startswith("buff."):
        io_ = pkg("pargcn_io")
        return io_.read_buffer_sizes(a) == io_.read_buffer_sizes(b) and ta.count("\n") == tb.count("\n") and len(ta) == len(tb)
    return ta == tb


@pytest.mark.parametrize("case", ["karate_k2", "karate_k3"])
def test_write_directory_reproduces_the_reference_tools_files(case, tmp_path):
    """Directories written by the reference's own `preprocess` + `GCN-HP` (committed fixtures): reading them and
    writing them again gives the same bytes (A.k: the same lines, the reference's row order is a hash order)."""
    io_ = pkg("pargcn_io")
    src = os.path.join(GOLDEN, "pargcn", case)
    prob = io_.load_directory(src)
    out = str(tmp_path / "out")
    io_.write_directory(out, prob["A"], prob["part"], prob["k"], prob["L"], prob["d"][1])
    names = sorted(os.listdir(src))
    assert names == sorted(os.listdir(out))
    for fn in names:
        assert _same_file(os.path.join(src, fn), os.path.join(out, fn), any_line_order=fn.startswith("A.")), fn
    again = io_.load_directory(out)
    assert (again["part"] == prob["part"]).all() and again["d"] == prob["d"]
    assert (abs(sp.csr_matrix(again["A"]) - sp.csr_matrix(prob["A"])) > 0).nnz == 0


def test_write_directory_larger_case_and_lossless_values(tmp_path):
    io_ = pkg("pargcn_io")
    with tarfile.open(os.path.join(GOLDEN, "pargcn", "gemat11p_k3.tar.gz")) as tf:
        tf.extractall(tmp_path)
    src = str(tmp_path / "out_gemat11p_k3")
    prob = io_.load_directory(src)
    out = str(tmp_path / "again")
    io_.write_directory(out, prob["A"], prob["part"], 3, prob["L"], prob["d"][1])
    for fn in sorted(os.listdir(src)):
        assert _same_file(os.path.join(src, fn), os.path.join(out, fn), any_line_order=fn.startswith("A.")), fn
    # "%.9g" keeps fp32 values exactly (the reference's %.2f does not)
    rng = np.random.default_rng(0)
    A = sp.random(60, 60, 0.1, random_state=1, dtype=np.float32, format="csr")
    A = (A + A.T).tocsr()
    part = rng.integers(0, 4, 60)
    io_.write_directory(str(tmp_path / "ll"), A, part, 4, 3, 8, value_format="%.9g")
    back = io_.load_directory(str(tmp_path / "ll"))
    assert (abs(sp.csr_matrix(back["A"]).astype(np.float32) - A) > 0).nnz == 0 and (back["part"] == part).all()
    # conn.k / buff.k equal what the engine's own partition derives (symmetric matrix)
    partition = pkg("partition")
    coo = A.tocoo()
    for p in range(4):
        pt = partition.build_partition(torch.from_numpy(coo.row.astype(np.int64)), torch.from_numpy(coo.col.astype(np.int64)),
                                       torch.from_numpy(coo.data), 60, torch.from_numpy(part), p, 4)
        conn, _ = back["conn"][p]
        mine = pt.send_map()
        assert sorted(conn) == sorted(q for q in mine if mine[q].numel())
        for q, ids in conn.items():
            assert np.array_equal(np.sort(ids), mine[q].numpy())


def test_partvec_writer_format_and_gz_reader(tmp_path):
    io_, partition = pkg("pargcn_io"), pkg("partition")
    ref = open(gpath("gemat11.mtx.3.hp")).read()            # written by the reference's GPU/hypergraph tool
    pv = list(map(int, ref.split()))
    io_.write_partvec(str(tmp_path / "x.3.hp"), pv)
    assert open(tmp_path / "x.3.hp").read() == ref
    import gzip
    with gzip.open(tmp_path / "x.3.hp.gz", "wt") as f:
        f.write(ref)
    assert partition.read_partvec(str(tmp_path / "x.3.hp.gz")) == pv == partition.read_partvec(str(tmp_path / "x.3.hp"))
    for k in (2, 4, 8):                                       # the committed part vectors of the mid workload
        for ext in ("hp", "gp"):
            v = partition.read_partvec(os.path.join(GOLDEN, "partvec", "mid.A.mtx.%d.%s.gz" % (k, ext)))
            assert len(v) == 131072 and min(v) == 0 and max(v) == k - 1


@pytest.mark.parametrize("min_entries,layer_min", [(8, 1), (64, 16), (512, 384)])
def test_strip_layout_round_trip(min_entries, layer_min):
    """build_strips: every stored entry ends up exactly once in the strip records or stays in the gather part;
    records are layers (<= 2 entries per row), padded slots point at the zero row, the first record of a piece
    always stages its panel."""
    partition, synth = pkg("partition"), pkg("synth")
    n, row, col, val = synth.make_graph(5000, 600000, seed=4)
    deg = torch.bincount(row, minlength=n)
    rank = torch.empty(n, dtype=torch.int64)
    rank[torch.argsort(-deg, stable=True)] = torch.arange(n)
    r, c = rank[row], rank[col]
    keep, st = partition.build_strips(r, c, val, n, n, min_entries=min_entries, layer_min=layer_min)
    assert st is not None
    rr, cc, vv = st.to_coo()
    k_all = torch.sort(r * n + c)
    k_got = torch.sort(torch.cat([rr * n + cc, (r * n + c)[keep]]))
    assert torch.equal(k_all.values, k_got.values)
    assert torch.equal(torch.cat([vv, val[keep]])[k_got.indices], val[k_all.indices])
    assert st.nnz == int((~keep).sum()) == int(st.rec_nnz.sum())
    off = st.pairs[:, :, 0]
    assert bool(((off == partition.STRIP_PAD_OFF) | ((off % 512 == 0) & (off >= 0) & (off < 128 * 512))).all())
    assert bool((st.pairs[:, :, 1][off == partition.STRIP_PAD_OFF] == 0).all())
    assert bool((st.rec_nnz >= layer_min).all())
    assert bool((st.rec[st.work[:, 1].long(), 1] == 0).all())
    # layers of a tile are consecutive records 0, 1, 2, ... with non-increasing counts
    tile = st.rec_tile_row.long() * (1 << 20) + st.rec[:, 0].long()
    same = tile[1:] == tile[:-1]
    assert bool((st.rec[1:, 3][same] == st.rec[:-1, 3][same] + 1).all())
    assert bool((st.rec_nnz[1:][same] <= st.rec_nnz[:-1][same]).all())
    # a record that starts a run of one panel names the panel of its piece's next run (-1: none), the others -1
    rec, work = st.rec.long(), st.work.long()
    for p in range(work.shape[0]):
        ks = list(range(int(work[p, 1]), int(work[p, 2])))
        starts = [k for k in ks if int(rec[k, 1]) == 0]
        for i, k in enumerate(starts):
            want = int(rec[starts[i + 1], 0]) if i + 1 < len(starts) else -1
            assert int(rec[k, 2]) == want
        assert all(int(rec[k, 2]) == -1 for k in ks if int(rec[k, 1]) == 1)
    # pieces tile the record list
    w = st.work[torch.argsort(st.work[:, 1])]
    assert int(w[0, 1]) == 0 and int(w[-1, 2]) == st.rec.shape[0] and bool((w[1:, 1] == w[:-1, 2]).all())


def test_csr_from_coo_with_strips_keeps_every_entry():
    partition = pkg("partition")
    rng = np.random.default_rng(3)
    D = (rng.random((1100, 900)) < 0.02) * rng.standard_normal((1100, 900))
    D[:256, :256] = (rng.random((256, 256)) < 0.5) * 1.0
    A = sp.csr_matrix(D.astype(np.float32))
    for nslices in (1, 8):
        h = partition.csr_from_scipy(A, nslices=nslices, core=True, dense3_tau=2.0, strip=True, strip_min=32)
        assert h.strip is not None and h.dense3 is None and h.core is None and h.nnz == A.nnz
        r, c, v = h.to_coo()
        B = sp.csr_matrix((v.numpy(), (r.numpy(), c.numpy())), shape=A.shape)
        assert (abs(B - A) > 0).nnz == 0
        assert h.row_flags is not None and int(h.row_flags.sum()) >= 1024


def test_sbm_generator_and_vertex_order():
    """The second stand-in has planted communities that label propagation recovers; R-MAT has none and keeps
    the degree order."""
    synth, partition = pkg("synth"), pkg("partition")
    n, nnz = 12000, 1200000
    n1, row, col, val = synth.make_graph(n, nnz, seed=1, generator="sbm")
    assert n1 == n and row.numel() == nnz + n
    k, kt = row * n + col, col * n + row
    assert torch.equal(torch.sort(k).values, torch.sort(kt).values)            # symmetric pattern
    assert int((row == col).sum()) == n                                         # self loops added by the normalisation
    rs = torch.zeros(n, dtype=torch.float64).index_add_(0, row, val.double() * torch.sqrt(torch.bincount(row, minlength=n).double())[col])
    assert torch.allclose(rs, torch.sqrt(torch.bincount(row, minlength=n).double()), rtol=1e-4)   # D^-1/2 (A+I) D^-1/2
    n2, r2, c2, v2 = synth.make_graph(n, nnz, seed=1, generator="sbm")
    assert torch.equal(row, r2) and torch.equal(col, c2) and torch.equal(val, v2)                # seeded
    gdeg = torch.bincount(row, minlength=n) + torch.bincount(col, minlength=n)
    go, gr, info = partition.vertex_order(row, col, n, gdeg, "auto")
    assert info["order"] == "community" and info["inside"] > 0.5 and info["largest_share"] < 0.25
    assert torch.equal(torch.sort(go).values, torch.arange(n))
    # vertices of one community are consecutive: far more entries near the diagonal than under the degree order
    near_c = float(((gr[row] - gr[col]).abs() < n // 8).double().mean())
    _, grd, _ = partition.vertex_order(row, col, n, gdeg, "degree")
    near_d = float(((grd[row] - grd[col]).abs() < n // 8).double().mean())
    assert near_c > near_d + 0.2
    # R-MAT: no communities -> auto falls back to the plain degree order
    _, rr, cr, _ = synth.make_graph(n, nnz, seed=1, generator="rmat")
    gd = torch.bincount(rr, minlength=n) + torch.bincount(cr, minlength=n)
    go2, _, info2 = partition.vertex_order(rr, cr, n, gd, "auto")
    assert info2["order"] == "degree" and torch.equal(go2, torch.argsort(-gd, stable=True))


def test_partition_is_order_invariant():
    """The community order only renumbers local rows and slab positions: maps and the operator stay the same."""
    synth, partition = pkg("synth"), pkg("partition")
    n, row, col, val = synth.make_graph(6000, 300000, seed=2, generator="sbm")
    pv = synth.random_partvec(n, 3, seed=4)
    A = sp.csr_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    old = partition.ORDER_MODE
    try:
        parts = {}
        for mode in ("degree", "community"):
            partition.ORDER_MODE = mode
            parts[mode] = partition.build_partition(row, col, val, n, pv, 1, 3)
    finally:
        partition.ORDER_MODE = old
    a, b = parts["degree"], parts["community"]
    assert b.order_info["order"] == "community" and a.order_info["order"] == "degree"
    assert torch.equal(torch.sort(a.owned).values, torch.sort(b.owned).values) and not torch.equal(a.owned, b.owned)
    for q in (0, 2):
        assert torch.equal(a.send_map()[q], b.send_map()[q]) and torch.equal(a.recv_map()[q], b.recv_map()[q])
    for pt in (a, b):                                         # A_loc in global numbering == the rank's diagonal block
        r, c, v = pt.A_loc.to_coo()
        own = pt.owned.numpy()
        B = sp.csr_matrix((v.numpy(), (own[r.numpy()], own[c.numpy()])), shape=(n, n))
        mask = np.zeros(n, bool); mask[own] = True
        ref = A[mask][:, mask]
        assert (abs(B[mask][:, mask] - ref) > 0).nnz == 0


def test_rank_local_generation_and_shards(tmp_path):
    """papers100M-scale path in miniature: every rank generates only its rows from the shared portable stream,
    the shards are written rank by rank (tools/make_shards.py), read back, and their union is the symmetric
    normalised matrix a single process generates."""
    import subprocess
    import sys
    from conftest import ROOT
    synth, ingest, partition = pkg("synth"), pkg("ingest"), pkg("partition")
    n, pairs, P = 3000, 40000, 4
    pv = synth.block_partvec(n, P)
    whole = synth.rmat_shard_keys(n, pairs, 0, torch.zeros(n, dtype=torch.int64), seed=0)
    parts = [synth.rmat_shard_keys(n, pairs, r, pv, seed=0) for r in range(P)]       # (the chunk size is part of the stream)
    assert torch.equal(torch.sort(torch.cat(parts)).values, whole)
    r_, c_ = whole // n, whole % n
    assert torch.equal(torch.sort(c_ * n + r_).values, whole) and int((r_ == c_).sum()) == n       # symmetric, A + I
    for r in range(P):
        assert bool((pv[parts[r] // n] == r).all())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_shards.py"), "--workload", "mid", "--scale", "0.02",
                          "--ranks", "4", "--out", str(tmp_path / "m")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    n2 = int(131072 * 0.02)
    rows, cols, vals = [], [], []
    for r in range(4):
        sh = ingest.read_shard(ingest.shard_path(str(tmp_path / "m"), r))
        assert sh["n"] == n2 and sh["nparts"] == 4 and np.array_equal(sh["rows"], np.nonzero(synth.block_partvec(n2, 4).numpy() == r)[0])
        a, b, c = ingest.shard_coo(sh)
        rows.append(a); cols.append(b); vals.append(c)
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n2, n2))
    assert (abs(A - A.T) > 1e-7).nnz == 0 and (A.diagonal() > 0).all()
    deg = np.diff(A.indptr)
    expect = 1.0 / np.sqrt(deg[A.tocoo().row] * deg[A.tocoo().col])
    np.testing.assert_allclose(A.tocoo().data, expect, rtol=1e-6)
    assert partition.read_partvec(str(tmp_path / "m") + ".4.bp") == synth.block_partvec(n2, 4).tolist()
