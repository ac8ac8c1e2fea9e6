"""Partition layout (integer work => bit-exact) against the reference's own
compute_communication_maps / get_partitiont_of_adjacency_matrix outputs."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
from scipy.io import mmread

from conftest import SPMM_CASES, golden, gpath, pkg, read_partvec


def _build(mtx, pv, rank, P):
    partition = pkg("partition")
    A = sp.coo_matrix(mmread(gpath(mtx)))
    part = read_partvec(gpath(pv))
    p = partition.build_partition(torch.from_numpy(A.row.astype(np.int64)),
                                  torch.from_numpy(A.col.astype(np.int64)),
                                  torch.from_numpy(A.data.astype(np.float32)), A.shape[0],
                                  torch.tensor(part), rank, P)
    return A, np.asarray(part), p


def _dense(csr, nrows=None):
    M = sp.csr_matrix((csr.val.numpy(), csr.col.numpy(), csr.rowptr.numpy()),
                      shape=(csr.nrows, csr.ncols)).toarray()
    if csr.row_map is not None:
        full = np.zeros((nrows, csr.ncols), np.float32)
        full[csr.row_map.numpy()] = M
        return full
    return M


@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES)
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
        assert p.send_off[r] == p.send_off[r + 1] and p.recv_off[r] == p.recv_off[r + 1]
        # send_idx are LOCAL ids of the send_map rows
        np.testing.assert_array_equal(p.owned.numpy()[p.send_idx.numpy()], p.send_global.numpy())


@pytest.mark.parametrize("name,mtx,pv,P", SPMM_CASES[:1] + SPMM_CASES[4:6])
def test_pieces_reassemble_the_row_block(name, mtx, pv, P):
    for r in range(P):
        A, part, p = _build(mtx, pv, r, P)
        Ad = A.toarray().astype(np.float32)
        own = p.owned.numpy()
        block = Ad[own]                                    # rows of rank r, global columns
        loc = _dense(p.A_loc)
        halo = _dense(p.A_halo, p.n_local)
        np.testing.assert_array_equal(loc, block[:, own])
        np.testing.assert_array_equal(halo, block[:, p.halo_global.numpy()])
        # nothing else in the block
        rest = block.copy()
        rest[:, own] = 0
        rest[:, p.halo_global.numpy()] = 0
        assert not rest.any()
        np.testing.assert_array_equal(_dense(p.A_loc_T), loc.T)
        np.testing.assert_array_equal(_dense(p.A_halo_T), halo.T)
        # column ids sorted inside every row
        for csr in (p.A_loc, p.A_halo, p.A_loc_T, p.A_halo_T):
            rp, c = csr.rowptr.numpy(), csr.col.numpy()
            for i in range(csr.nrows):
                assert (np.diff(c[rp[i]:rp[i + 1]]) >= 0).all()


def test_edge_cases():
    partition = pkg("partition")
    z = torch.zeros(0, dtype=torch.int64)
    # a rank that owns nothing, and an empty matrix
    p = partition.build_partition(torch.tensor([0, 1]), torch.tensor([1, 0]), torch.ones(2), 2,
                                  torch.tensor([0, 0]), 1, 2)
    assert p.n_local == 0 and p.n_halo == 0 and p.n_send == 0 and p.A_loc.nnz == 0
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
