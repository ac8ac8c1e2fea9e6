"""The host-built launch plans, executed on the CPU (tests/plan_interpreter.py): every structure the HIP kernels read --
gather tasks and their XCD segments, fix records, strip records (layers, pad slots, panel flags, the next-run panel),
LDS-core tiles, MFMA tiles in A-operand order, the per-row slot lists of the fix-up, row maps of the halo blocks -- is
decoded the way include/pgcn_hip.h specifies and must reproduce A . B to float64 round-off.  Host logic only: what a
wrong offset in partition.py / kernels.py / pgcn_spmm_plan_host would otherwise show on a GPU first."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.io import mmread

from conftest import gpath, pkg, read_partvec
from plan_interpreter import HostPlanner, run_plan

TOL = 1e-12


def _power_law_block(n=3000, nnz=300000, seed=4):
    """A degree-sorted synthetic block: dense corner, sparse tail (what every rank's local block looks like)."""
    synth = pkg("synth")
    n, row, col, val = synth.make_graph(n, nnz, seed=seed)
    deg = torch.bincount(row, minlength=n)
    rank = torch.empty(n, dtype=torch.int64)
    rank[torch.argsort(-deg, stable=True)] = torch.arange(n)
    r, c = rank[row], rank[col]
    A = sp.csr_matrix((val.numpy().astype(np.float64), (r.numpy(), c.numpy())), shape=(n, n))
    return n, r, c, val, A


VARIANTS = {
    # name: (csr_from_coo arguments, planner arguments, parts that must be present)
    "strips+bf16x3": (dict(nslices=8, core=True, strip=True, strip_min=64, dense3_tau=0.12), {}, ("strip", "dense3")),
    "strips_only": (dict(nslices=8, core=True, strip=True, strip_min=64, dense3_tau=2.0), {}, ("strip",)),
    "core_only": (dict(nslices=8, core=True, strip=False, tau=0.05, emax=5000, dense3_tau=2.0), {}, ("core",)),
    "gather_sliced": (dict(nslices=8, core=False), {}, ()),
    "gather_sliced_short_tasks": (dict(nslices=8, core=False), dict(chunk=64, small_row=16, adaptive_chunk=False), ()),
    "gather_unsliced_long_rows": (dict(nslices=1, core=False), dict(chunk=128, adaptive_chunk=False), ()),
    "gather_column_groups": (dict(nslices=8, core=False, ngroups=4), {}, ()),
    "strips+bf16x3+short_rows": (dict(nslices=8, core=True, strip=True, strip_min=64, dense3_tau=0.12), dict(small_row=8, adaptive_chunk=False),
                                 ("strip", "dense3")),
    # r06: the 512 x 128 grid of the bf16 blocks restarts at every band start (communities of the vertex order): blocks at odd origins
    "strips+bf16x3+bands": (dict(nslices=8, core=True, strip=True, strip_min=64, dense3_tau=0.12, row_bands=[0, 300, 1250, 2100],
                                 col_bands=[0, 300, 1250, 2100]), {}, ("strip", "dense3")),
    "range_slices": (dict(core=True, strip=True, strip_min=64, dense3_tau=0.15, slice_bounds=[0, 40, 100, 250, 600, 1100, 1700, 2400, 3000]), {},
                     ("strip", "dense3")),
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_launch_group_plan_reproduces_the_product(name):
    partition = pkg("partition")
    kw, pk, parts = VARIANTS[name]
    n, r, c, val, A = _power_law_block()
    kw = {k: (torch.tensor(x) if k.endswith("_bands") else x) for k, x in kw.items()}
    h = partition.csr_from_coo(r, c, val, n, n, **kw)
    if "row_bands" in kw:            # every block starts at a band start + a multiple of the block size, and none crosses a band
        d3, bands = h.dense3, kw["row_bands"].tolist() + [n]
        for r0, c0 in zip(d3.blk_row0.tolist(), d3.blk_col0.tolist()):
            br = max(b for b in bands[:-1] if b <= r0)
            bc = max(b for b in bands[:-1] if b <= c0)
            assert (r0 - br) % 512 == 0 and (c0 - bc) % 128 == 0
        rr, cc, _ = d3.coo
        for b in range(d3.blk_row.numel()):
            pass
        assert any(x % 512 for x in d3.blk_row0.tolist()) and any(x % 128 for x in d3.blk_col0.tolist())
        assert bool((d3.piece_rows <= 512).all()) and bool((d3.piece_rows > 0).all())
    for p in ("strip", "core", "dense3"):
        assert (getattr(h, p) is not None) == (p in parts), "variant %s: part %s" % (name, p)
    assert h.nnz == A.nnz
    d = HostPlanner(**pk).prepare(h)
    rng = np.random.default_rng(1)
    B = rng.standard_normal((n, 6))
    ref = A @ B
    slice_of = None
    if "slice_bounds" in kw:        # range slices: a task's slice is the column range it falls in, not col % 8
        inner = np.asarray(kw["slice_bounds"][1:-1])
        slice_of = lambda cols: np.searchsorted(inner, cols, side="right")      # noqa: E731
    C, info = run_plan(d, B, slice_of=slice_of)
    assert not np.isnan(C).any(), "rows nobody wrote"
    assert np.abs(C - ref).max() < TOL
    assert info["entries_gather"] == h.col.numel()
    assert info["entries_strip"] == (h.strip.nnz if h.strip is not None else 0)
    assert info["entries_core"] == (h.core.nnz if h.core is not None else 0)
    if "short_tasks" in name or "long_rows" in name:
        assert d.nfix > 100                                   # many rows are cut into several tasks
    # accumulate: C0 + A . B through the same plan
    C0 = rng.standard_normal((n, 6))
    C2, _ = run_plan(d, B, C0=C0, accumulate=True, slice_of=slice_of)
    assert np.abs(C2 - (C0 + ref)).max() < TOL
    # pattern-only upload (the unpack matrices of the backward exchange): all values one
    if not parts:
        dp = HostPlanner(**pk).prepare(h, pattern_only=True)
        Cp, _ = run_plan(dp, B)
        ones = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape)
        assert np.abs(Cp - ones @ B).max() < TOL


def test_row_subset_block_writes_only_its_rows():
    """compact_rows: CSR over the rows that have entries + row_map to the output rows (the row-subset form of
    pgcn_spmm_csr_plan_f32); rows outside the map are never touched."""
    partition = pkg("partition")
    n, r, c, val, A = _power_law_block(n=2000, nnz=120000, seed=9)
    keep = (r % 3 == 1)
    h = partition.csr_from_coo(r[keep], c[keep], val[keep], n, n, compact_rows=True, nslices=8)
    assert h.row_map is not None and h.nrows == int(torch.unique(r[keep]).numel()) < n
    d = HostPlanner(chunk=64, adaptive_chunk=False).prepare(h)
    B = np.random.default_rng(3).standard_normal((n, 4))
    C0 = np.random.default_rng(4).standard_normal((n, 4))
    sub = sp.csr_matrix((val[keep].numpy().astype(np.float64), (r[keep].numpy(), c[keep].numpy())), shape=(n, n))
    C, _ = run_plan(d, B, C0=C0, accumulate=True)
    assert np.abs(C - (C0 + sub @ B)).max() < TOL
    C, _ = run_plan(d, B, C0=C0)                            # overwrite mode: mapped rows replaced, the others untouched
    rows = h.row_map.numpy()
    other = np.setdiff1d(np.arange(n), rows)
    assert np.abs(C[rows] - (sub @ B)[rows]).max() < TOL and np.array_equal(C[other], C0[other])


def test_ragged_block_and_the_last_panel_window():
    """A 700 x 520 block: the last strip panel is the window [392, 520) (no staged row lies past the operand), the last
    MFMA panel is partial, the last tile row has 60 rows."""
    partition = pkg("partition")
    rng = np.random.default_rng(4)
    n, m = 700, 520
    D = (rng.random((n, m)) < 0.02).astype(np.float64)
    D[:256, :256] = rng.random((256, 256)) < 0.6
    D[256:384, :128] = rng.random((128, 128)) < 0.12
    D[640:, 384:] = rng.random((60, 136)) < 0.5
    D[100:600, 400:520] += rng.random((500, 120)) < 0.08            # entries of the windowed last panel
    D *= rng.standard_normal((n, m))
    A = sp.coo_matrix(D)
    B = rng.standard_normal((m, 5))
    for kw in (dict(strip=True, strip_min=32, dense3_tau=0.2), dict(strip=True, strip_min=32, dense3_tau=2.0),
               dict(strip=False, tau=0.05, emax=3000, dense3_tau=2.0)):
        h = partition.csr_from_scipy(A, nslices=1, core=True, **kw)
        if kw["strip"]:
            assert h.strip is not None and int(h.strip.rec[:, 0].max()) == 4            # panel 4 = columns 392..519
            assert partition.strip_panel_base(4, m) == m - 128
        d = HostPlanner().prepare(h)
        C, info = run_plan(d, B)
        assert not np.isnan(C).any()
        assert np.abs(C - D.astype(np.float32).astype(np.float64) @ B).max() < TOL
        rr, cc, vv = h.to_coo()
        back = sp.coo_matrix((vv.numpy(), (rr.numpy(), cc.numpy())), shape=(n, m)).toarray()
        np.testing.assert_array_equal(back, D.astype(np.float32))


def test_duplicate_coordinates_deeper_than_a_tile_keep_their_sum():
    """An uncoalesced COO keeps duplicate coordinates as separate stored entries (PGCN.py:63 sums them).  A row can
    then hold more than 128 entries in one 128-column panel = more than the 64 layers a strip tile has: the surplus
    stays in the gather part and nothing is lost (the r02 advisor's finding on build_strips)."""
    partition = pkg("partition")
    rng = np.random.default_rng(5)
    n = 1024
    base = sp.random(n, n, density=0.03, random_state=rng, data_rvs=lambda k: rng.uniform(-1, 1, k)).tocoo()
    dup_r = np.repeat(np.arange(200), 150)                              # rows 0..199: 150 extra entries each inside panel 0
    dup_c = rng.integers(0, 128, dup_r.size)                            # = 75+ layers of 400 entries in strip tile (0, 0)
    r = torch.from_numpy(np.concatenate([base.row, dup_r]).astype(np.int64))
    c = torch.from_numpy(np.concatenate([base.col, dup_c]).astype(np.int64))
    v = torch.from_numpy(np.concatenate([base.data, rng.uniform(-1, 1, dup_r.size)]).astype(np.float32))
    A = sp.coo_matrix((v.numpy().astype(np.float64), (r.numpy(), c.numpy())), shape=(n, n)).tocsr()      # sums duplicates
    h = partition.csr_from_coo(r, c, v, n, n, nslices=1, core=True, strip=True, strip_min=32, dense3_tau=2.0)
    assert h.strip is not None and h.nnz == r.numel()
    assert int(h.strip.rec[:, 3].max()) <= 63
    sr, sc, _ = h.strip.to_coo()
    in_tile = (sr < 200) & (sc < 128)
    assert int(in_tile.sum()) == 200 * 128                              # 64 layers x 2 slots per row; the rest stays behind
    left = (h.rowptr[1:] - h.rowptr[:-1])[:200]                          # ... in the gather part: 150 + a few - 128 per row
    assert h.core is None and int((left >= 150 - 128).sum()) == 200
    B = rng.standard_normal((n, 4))
    C, _ = run_plan(HostPlanner().prepare(h), B)
    assert np.abs(C - A @ B).max() < 1e-11


@pytest.mark.parametrize("mtx,pv,P", [("gemat11.mtx", "gemat11.mtx.3.hp", 3), ("karate.mtx", "karate.mtx.2.rp", 2)])
def test_partition_blocks_forward_and_backward_through_their_plans(mtx, pv, P):
    """Every rank's local block, halo blocks (one per exchange round), transposes and unpack patterns as
    build_partition hands them to the engine, executed from their plans: forward = (A . H)[owned rows], backward =
    (A^T . G)[owned rows] with the partial rows returned to their owners slab position by slab position."""
    partition = pkg("partition")
    A = sp.coo_matrix(mmread(gpath(mtx))).astype(np.float32)
    n = A.shape[0]
    part = torch.tensor(read_partvec(gpath(pv)))
    row, col, val = (torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64)), torch.from_numpy(A.data))
    A64 = sp.csr_matrix(A).astype(np.float64)
    rng = np.random.default_rng(2)
    H, G = rng.standard_normal((n, 5)), rng.standard_normal((n, 5))
    AH, ATG = A64 @ H, A64.T @ G
    K = HostPlanner()
    parts = [partition.build_partition(row, col, val, n, part, p, P) for p in range(P)]
    partials = []
    for p, pt in enumerate(parts):
        own, hg = pt.owned.numpy(), pt.halo_global.numpy()
        C, _ = run_plan(K.prepare(pt.A_loc), H[own])
        C = np.nan_to_num(C, nan=0.0) if pt.A_loc.nnz == 0 else C
        for a in pt.A_halo:
            d = K.prepare(a)
            C, _ = run_plan(d, H[hg], C0=C, accumulate=True)
        assert np.abs(C - AH[own]).max() < TOL, "forward, rank %d" % p
        # backward, this rank's half: partial rows for the halo (round by round), dH of the local block
        partial = np.zeros((pt.n_halo, 5))
        for r_, at in enumerate(pt.A_halo_T):
            b0, b1 = pt.round_recv_off[r_][0], pt.round_recv_off[r_][-1]
            if b1 > b0:
                out, _ = run_plan(K.prepare(at), G[own])
                partial[b0:b1] = np.nan_to_num(out, nan=0.0)
        partials.append(partial)
    for q, pt in enumerate(parts):
        own = pt.owned.numpy()
        dH, _ = run_plan(K.prepare(pt.A_loc_T), G[own])
        # the wire: slab position j of rank q's send side receives peer send_owner[j]'s partial row for send_global[j]
        back = np.zeros((pt.n_send, 5))
        sg, so = pt.send_global.numpy(), pt.send_owner.numpy()
        for j in range(pt.n_send):
            peer = parts[int(so[j])]
            (i,) = np.nonzero(peer.halo_global.numpy() == sg[j])
            assert i.size == 1
            back[j] = partials[int(so[j])][int(i[0])]
        for u in pt.unpack:
            dH, _ = run_plan(K.prepare(u, pattern_only=True), back, C0=dH, accumulate=True)
        assert np.abs(dH - ATG[own]).max() < TOL, "backward, rank %d" % q


def test_attention_structures_of_a_rank():
    """What the GAT kernels read besides the SpMM plan (gat.build_gat_graph + HipKernels.prepare_gat): the forward
    pattern over [local ; halo] columns, its transpose with the entry permutation, the wave / workgroup row lists
    (every row in exactly one) and the per-row offsets of the eight col % 8 slices."""
    partition, gat = pkg("partition"), pkg("gat")
    A = sp.coo_matrix(mmread(gpath("gemat11p.A.mtx"))).astype(np.float32)
    n = A.shape[0]
    part = torch.tensor(read_partvec(gpath("gemat11.mtx.3.hp")))
    pt = partition.build_partition(torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64)),
                                   torch.from_numpy(A.data), n, part, 1, 3)
    g = gat.build_gat_graph(pt)
    K = HostPlanner()
    for h, wave, block in ((g.fwd, g.fwd_wave, g.fwd_block), (g.bwd, g.bwd_wave, g.bwd_block)):
        d = K.prepare_gat(h, wave, block)
        listed = np.sort(np.concatenate([d.rows_wave.numpy(), d.rows_block.numpy()]))
        ln = (h.rowptr[1:] - h.rowptr[:-1]).numpy()
        assert np.array_equal(listed, np.nonzero(ln > 0)[0]) or np.array_equal(listed, np.arange(h.nrows))
        if d.slice_off is not None:
            off, col, rp = d.slice_off.numpy().astype(np.int64), h.col.numpy().astype(np.int64), h.rowptr.numpy()
            assert np.array_equal(off[:, 8], ln) and (np.diff(off, axis=1) >= 0).all()
            for r in np.argsort(-ln)[:50]:
                for s_ in range(8):
                    assert (col[rp[r] + off[r, s_]:rp[r] + off[r, s_ + 1]] % 8 == s_).all()
            assert np.array_equal(np.sort(d.rows_all.numpy()), np.arange(h.nrows))
            assert (np.diff(ln[d.rows_all.numpy().astype(np.int64)]) <= 0).all()              # longest first
        # the pattern SpMM plan of the structure reproduces the product with all values one
        B = np.random.default_rng(0).standard_normal((h.ncols, 3))
        C, _ = run_plan(d, B)
        ones = sp.csr_matrix((np.ones(h.col.numel()), h.col.numpy(), h.rowptr.numpy()), shape=(h.nrows, h.ncols))
        assert np.abs(np.nan_to_num(C) - ones @ B).max() < TOL
    # entry p of the transposed structure is entry perm[p] of the forward one
    fr, fc, _ = g.fwd.to_coo()
    br, bc, _ = g.bwd.to_coo()
    perm = g.perm.numpy()
    assert np.array_equal(np.sort(perm), np.arange(perm.size))
    assert np.array_equal(br.numpy(), fc.numpy()[perm]) and np.array_equal(bc.numpy(), fr.numpy()[perm])


def test_default_tuning_on_a_dense_graph_reaches_every_tile_path(monkeypatch):
    """The Reddit shape at 1/8 of its vertices with its average degree (492 stored entries per row) under the SHIPPED
    tuning (except that this small matrix may keep its 150 bf16 blocks: the shipped minimum is 400), through
    build_partition as bench.py calls it: one rank gets strip tiles + bf16 blocks (and the same for the pre-built
    transpose), a rank of four gets a gather-only local block and a tiled halo block.
    Forward through the plans = (A . H)[owned rows], transposed block = A^T . H."""
    partition, synth = pkg("partition"), pkg("synth")
    monkeypatch.setattr(partition, "DENSE3_MIN_BLOCKS", 0)
    n, row, col, val = synth.make_graph(29120, 14326986, seed=0)
    A = sp.csr_matrix((val.numpy().astype(np.float64), (row.numpy(), col.numpy())), shape=(n, n))
    H = np.random.default_rng(0).standard_normal((n, 2))
    AH, ATH = A @ H, A.T @ H
    K = HostPlanner()
    pt = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64), 0, 1)
    own = pt.owned.numpy()
    for blk, ref in ((pt.A_loc, AH), (pt.A_loc_T, ATH)):
        assert blk.strip is not None and blk.dense3 is not None and blk.col.numel() > 0 and blk.nnz == A.nnz
        assert blk.strip.nnz > 0.2 * A.nnz and blk.dense3.nnz > 0.2 * A.nnz
        C, info = run_plan(K.prepare(blk), H[own])
        assert np.abs(C - ref[own]).max() < TOL
    pt = partition.build_partition(row, col, val, n, synth.random_partvec(n, 4, seed=0), 1, 4)
    own, hg = pt.owned.numpy(), pt.halo_global.numpy()
    assert pt.A_loc.strip is None and pt.A_loc.core is None and pt.A_loc.dense3 is None          # a small block: gather only
    assert any(a.core is not None or a.dense3 is not None or a.strip is not None for a in pt.A_halo)
    C, _ = run_plan(K.prepare(pt.A_loc), H[own])
    for a in pt.A_halo:
        C, _ = run_plan(K.prepare(a), H[hg], C0=C, accumulate=True)
    assert np.abs(C - AH[own]).max() < TOL


@pytest.mark.parametrize("tuning,parts1,parts3,rounds3", [
    ("core_min_nnz=0,core_min_frac=0,strip_min_records=0,dense3_min_blocks=0,exchange_rounds=3", "s3", "s3", 3),   # strips + bf16 blocks on a shard too
    ("dense_bf16x3=0,core_min_nnz=0,core_min_frac=0,strip_min_records=0", "s", "s", 2),          # no bf16 blocks: their entries go to the strips
    ("strip=0,dense_bf16x3=0,core_min_nnz=0,core_emax=500,exchange_rounds=1", "c", "c", 1),      # LDS core with short pieces
    ("tiles=0,slices=1,spmm_chunk=64,spmm_adaptive_chunk=0", "-", "-", 2),                       # gather only, unsliced, short tasks
])
def test_non_default_tunings_keep_the_plans_exact(tuning, parts1, parts3, rounds3):
    """PGCN_TUNING is the one switchboard of the path (tuning.py); whatever it selects, the plans must still be the
    product.  One interpreter per tuning (the variable is read once at import): tests/_plan_tuning_child.py."""
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_plan_tuning_child.py")
    out = subprocess.run([sys.executable, child], env=dict(os.environ, PGCN_TUNING=tuning), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = re.findall(r"P=(\d) parts=(\S+) rounds=(\d+) err=(\S+)", out.stdout)
    assert [(g[0], g[1]) for g in got] == [("1", parts1), ("3", parts3)] and int(got[1][2]) == rounds3
    assert all(float(g[3]) < TOL for g in got)
