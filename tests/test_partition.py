"""Partition layout (integer work => bit-exact) against the reference's own
compute_communication_maps / get_partitiont_of_adjacency_matrix outputs."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.io import mmread

from conftest import SPMM_CASES, SPMM_CASES_MORE, golden, gpath, pkg, read_partvec


def _build(mtx, pv, rank, P, rounds=None):
    partition = pkg("partition")
    A = sp.coo_matrix(mmread(gpath(mtx)))
    part = read_partvec(gpath(pv))
    p = partition.build_partition(torch.from_numpy(A.row.astype(np.int64)),
                                  torch.from_numpy(A.col.astype(np.int64)),
                                  torch.from_numpy(A.data.astype(np.float32)), A.shape[0],
                                  torch.tensor(part), rank, P, rounds=rounds)
    return A, np.asarray(part), p


def _sum_dense(csrs, nrows, ncols, row_base=None):
    out = np.zeros((nrows, ncols), np.float32)
    for k, c in enumerate(csrs):
        d = _dense(c, nrows if row_base is None else None)
        if row_base is None:
            out += d
        else:
            out[row_base[k]:row_base[k] + d.shape[0]] += d
    return out


def _dense(csr, nrows=None):
    r, c, v = csr.to_coo()                      # gather part + dense core, output-row numbering
    R = csr.nrows if csr.row_map is None else nrows
    return sp.csr_matrix((v.numpy(), (r.numpy(), c.numpy())), shape=(R, csr.ncols)).toarray()


@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES + SPMM_CASES_MORE)
def test_maps_and_counts_match_reference(name, mtx, pv, P):
    arrays, meta = golden(name)
    for r in range(P):
        A, part, p = _build(mtx, pv, r, P)
        m = meta["ranks"][r]
        assert p.n_local == m["n_local"] and p.nnz_local == m["nnz_local"]
        smap, rmap = p.send_map(), p.recv_map()
        assert sorted(smap) == [q for q in range(P) if q != r]
        for q in smap:
            np.testing.assert_array_equal(smap[q].numpy(), arrays["send_%d_%d" % (r, q)])
            np.testing.assert_array_equal(rmap[q].numpy(), arrays["recv_%d_%d" % (r, q)])
        for off in p.round_send_off + p.round_recv_off:          # own-rank segments are empty
            assert off[r] == off[r + 1]
        assert not (p.send_owner == r).any() and not (p.halo_owner == r).any()
        # send_idx are LOCAL ids of the send slab rows
        np.testing.assert_array_equal(p.owned.numpy()[p.send_idx.numpy()], p.send_global.numpy())
        # the rounds tile both slabs without gaps, peers ascending inside a round
        for offs, owner, total in ((p.round_send_off, p.send_owner, p.n_send), (p.round_recv_off, p.halo_owner, p.n_halo)):
            pos = 0
            for off in offs:
                assert off[0] == pos and all(a <= b for a, b in zip(off, off[1:]))
                for q in range(P):
                    assert (owner.numpy()[off[q]:off[q + 1]] == q).all()
                pos = off[-1]
            assert pos == total


# (1, 2 and 3 exchange rounds on the small graph; the 4 929-vertex ones -- 2.5 s of partition build per rank -- with the
#  shipped two rounds and one other count each)
@pytest.mark.parametrize("name,mtx,pv,P,rounds",
                         [SPMM_CASES[0] + (r,) for r in (1, 2, 3)] + [SPMM_CASES[4] + (r,) for r in (2, 3)]
                         + [SPMM_CASES[5] + (r,) for r in (1, 2)])
def test_pieces_reassemble_the_row_block(name, mtx, pv, P, rounds):
    for r in range(P):
        A, part, p = _build(mtx, pv, r, P, rounds)
        assert p.rounds == rounds
        Ad = A.toarray().astype(np.float32)
        own = p.owned.numpy()
        block = Ad[own]                                    # rows of rank r, global columns
        loc = _dense(p.A_loc)
        halo = _sum_dense(p.A_halo, p.n_local, p.n_halo)
        # round k's matrix only touches round k's sub-slab
        for k, a in enumerate(p.A_halo):
            cols = a.to_coo()[1].numpy()
            assert ((cols >= p.round_recv_off[k][0]) & (cols < p.round_recv_off[k][-1])).all()
        np.testing.assert_array_equal(loc, block[:, own])
        np.testing.assert_array_equal(halo, block[:, p.halo_global.numpy()])
        # nothing else in the block
        rest = block.copy()
        rest[:, own] = 0
        rest[:, p.halo_global.numpy()] = 0
        assert not rest.any()
        np.testing.assert_array_equal(_dense(p.A_loc_T), loc.T)
        np.testing.assert_array_equal(
            _sum_dense(p.A_halo_T, p.n_halo, p.n_local, [o[0] for o in p.round_recv_off]), halo.T)
        # inside every row: grouped by slice (col % nslices), ascending inside a slice
        for csr in [p.A_loc, p.A_loc_T] + p.A_halo + p.A_halo_T:
            rp, c = csr.rowptr.numpy(), csr.col.numpy().astype(np.int64)
            key = (c % csr.nslices) * (csr.ncols + 1) + c
            for i in range(csr.nrows):
                assert (np.diff(key[rp[i]:rp[i + 1]]) >= 0).all()
        # local numbering and slab orders = decreasing GLOBAL degree (stored entries, row + column)
        gd = np.bincount(A.row, minlength=A.shape[0]) + np.bincount(A.col, minlength=A.shape[0])
        assert (np.diff(gd[own]) <= 0).all()
        for k in range(p.rounds):
            for q in range(P):
                a, b = p.round_recv_off[k][q], p.round_recv_off[k][q + 1]
                assert (np.diff(gd[p.halo_global.numpy()[a:b]]) <= 0).all()
                a, b = p.round_send_off[k][q], p.round_send_off[k][q + 1]
                assert (np.diff(gd[p.send_global.numpy()[a:b]]) <= 0).all()


def test_edge_cases():
    partition = pkg("partition")
    z = torch.zeros(0, dtype=torch.int64)
    # a rank that owns nothing, and an empty matrix
    p = partition.build_partition(torch.tensor([0, 1]), torch.tensor([1, 0]), torch.ones(2), 2,
                                  torch.tensor([0, 0]), 1, 2)
    assert p.n_local == 0 and p.n_halo == 0 and p.n_send == 0 and p.A_loc.nnz == 0 and p.rounds == 2
    p = partition.build_partition(z, z, torch.zeros(0), 3, torch.tensor([0, 1, 0]), 0, 2)
    assert p.n_local == 2 and p.A_loc.rowptr.tolist() == [0, 0, 0]
    with pytest.raises(ValueError):
        partition.build_partition(z, z, torch.zeros(0), 3, torch.tensor([0, 1]), 0, 2)
    with pytest.raises(ValueError):
        partition.build_partition(z, z, torch.zeros(0), 3, torch.tensor([0, 2, 0]), 0, 2)
    # duplicates in the COO are kept (summed by the SpMM like an uncoalesced COO)
    p = partition.build_partition(torch.tensor([0, 0]), torch.tensor([1, 1]), torch.tensor([1., 2.]), 2,
                                  torch.tensor([0, 0]), 0, 1)
    assert p.A_loc.nnz == 2


def test_dense_core_split_roundtrip():
    """Entries of dense 128x128 tiles move to the LDS-tiled layout; nothing is lost or duplicated."""
    partition, synth, kernels = pkg("partition"), pkg("synth"), pkg("kernels")
    n, row, col, val = synth.make_graph(3000, 300000, seed=4)
    deg = torch.bincount(row, minlength=n)
    rank = torch.empty(n, dtype=torch.int64)
    rank[torch.argsort(-deg, stable=True)] = torch.arange(n)
    r, c = rank[row], rank[col]
    h = partition.csr_from_coo(r, c, val, n, n, nslices=8, core=True, tau=0.05, emax=5000, strip=False, dense3_tau=2.0)
    assert h.core is not None and h.core.nnz > 0.2 * r.numel() and h.core.npieces > h.core.tile_row.unique().numel()
    assert h.nnz == r.numel()
    A = sp.csr_matrix((val.numpy(), (r.numpy(), c.numpy())), shape=(n, n))
    rr, cc, vv = h.to_coo()
    B = sp.csr_matrix((vv.numpy(), (rr.numpy(), cc.numpy())), shape=(n, n))
    assert abs(A - B).max() == 0 and B.nnz == A.nnz
    co = h.core
    # every dense tile really is dense, every piece stays inside one row tile, pieces cover all tiles once
    tot = (co.seg_off[:, -1]).numpy()
    assert (tot >= int(0.05 * 128 * 128)).all()
    w = co.work.numpy()
    cover = np.zeros(len(tot), int)
    for tr, kb, ke, sb in w:
        assert (co.tile_row.numpy()[kb:ke] == tr).all()
        cover[kb:ke] += 1
    assert (cover == 1).all() and sorted(w[:, 3].tolist()) == [128 * i for i in range(len(w))]
    edges = [tot[kb:ke].sum() for _, kb, ke, _ in w]
    assert all(a >= b for a, b in zip(edges, edges[1:]))
    assert (co.ccol.numpy() < 128).all() and (co.ccol.numpy() >= 0).all()
    # flagged rows = rows of the core row tiles; the gather plan never writes them directly
    flags = h.row_flags.numpy()
    assert set(np.nonzero(flags)[0] // 128) == set(co.tile_row.numpy().tolist())
    tasks, fix, nslots, seg = kernels.build_plan(h.rowptr.numpy(), 1024, h.slice_cnt.numpy(), 96, row_flags=flags)
    dst = tasks[:, 3]
    direct_rows = ~dst[dst < 0]
    assert not flags[direct_rows].any()
    assert set(fix[:, 0].tolist()) >= set(np.nonzero(flags & (np.diff(h.rowptr.numpy()) > 0))[0].tolist())


def test_synthetic_graph_is_normalised_symmetric():
    synth = pkg("synth")
    from oracle import oracle
    n, row, col, val = synth.make_graph(500, 6000, seed=1)
    assert row.numel() == 6000 + 500
    A = sp.csr_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    assert abs(A - A.T).max() < 1e-7
    pat = sp.csr_matrix((np.ones(row.numel()), (row.numpy(), col.numpy())), shape=(n, n))
    ref = oracle.normalize_adjacency(pat)      # restatement of preprocess/GrB-GNN-IDG.py
    assert abs(A - ref).max() < 1e-6
    n2, r2, c2, v2 = synth.make_graph(500, 6000, seed=1)
    assert torch.equal(row, r2) and torch.equal(val, v2)




def test_single_rank_symmetric_block_shares_its_transpose():
    """r06: at P = 1 a matrix that equals its transpose entry for entry (bit-compared, duplicates included) builds ONE set of
    structures (A_loc_T is A_loc); an unsymmetric one, or one whose values differ in the last bit, builds both."""
    partition, synth = pkg("partition"), pkg("synth")
    n, row, col, val = synth.make_graph(900, 20000, seed=3)
    pv = torch.zeros(n, dtype=torch.int64)
    p = partition.build_partition(row, col, val, n, pv, 0, 1)
    assert p.A_loc_T is p.A_loc
    np.testing.assert_array_equal(_dense(p.A_loc), _dense(p.A_loc).T)
    v2 = val.clone()
    k = int(torch.nonzero(row != col)[0])
    v2[k] = torch.nextafter(v2[k], torch.tensor(2.0))                 # one entry off by an ulp: no longer its own transpose
    q = partition.build_partition(row, col, v2, n, pv, 0, 1)
    assert q.A_loc_T is not q.A_loc
    np.testing.assert_array_equal(_dense(q.A_loc_T), _dense(q.A_loc).T)
    keep = ~((row == row[k]) & (col == col[k]))                       # an unsymmetric pattern
    u = partition.build_partition(row[keep], col[keep], val[keep], n, pv, 0, 1)
    assert u.A_loc_T is not u.A_loc
    # two ranks: never shared (the local block of a rank is square but its transpose is built with the halo logic)
    p2 = partition.build_partition(row, col, val, n, synth.random_partvec(n, 2, seed=0), 0, 2)
    assert p2.A_loc_T is not p2.A_loc
    # duplicates stored in another order on the two sides still count as symmetric
    r = torch.tensor([0, 1, 0, 1, 0, 1]); c = torch.tensor([1, 0, 1, 0, 0, 1]); v = torch.tensor([1.0, 2.0, 2.0, 1.0, 5.0, 6.0])
    assert partition._coo_is_symmetric(r, c, v, 2)
    assert not partition._coo_is_symmetric(r, c, torch.tensor([1.0, 2.0, 2.0, 3.0, 5.0, 6.0]), 2)
