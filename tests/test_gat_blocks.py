"""The dense blocks of the attention pattern on the bf16 matrix cores (csrc/pgcn_gat_blocks.hip, r06).
CPU: the pattern bits against partition.dense3_index, position by position.  GPU: gather part + block part against the gather kernels
over the WHOLE pattern (pgcn_spmm_heads_forward2_f32 / _grad_f32, themselves held to the numpy oracle in test_gat_gpu.py) and against
the float64 oracle of /root/reference/GPU/PGAT.py:138-151 directly; the engine with and without the blocks."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import pkg, rel_err
from oracle import oracle


def _corner_graph(n, m, seed, fill=0.3, rows=700, cols=300):
    """A sparse n x m pattern with a dense top-left corner (what the degree order makes of a power-law graph), a hub row,
    a hub column and an empty row."""
    rng = np.random.default_rng(seed)
    A = sp.random(n, m, density=0.01, random_state=seed, format="lil")
    R, C = min(rows, n), min(cols, m)
    A[:R, :C] = (rng.random((R, C)) < fill).astype(np.float32)
    A[3, :] = 1
    A[:, 5] = 1
    A[7, :] = 0
    A = sp.csr_matrix(A)
    A.data[:] = 1
    A.eliminate_zeros()
    A.sort_indices()
    return A, rng


def _coords(A):
    A = sp.coo_matrix(A)
    return torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64))


def test_pattern_bits_follow_the_operand_order():
    partition, kernels = pkg("partition"), pkg("kernels")
    A, _ = _corner_graph(1100, 700, 3)
    r, c = _coords(A)
    keep, h3 = partition.split_dense3(r, c, torch.ones(r.numel()), 1100, 700, 0.06)
    assert h3 is not None and 0 < h3.nnz < A.nnz and int((~keep).sum()) == h3.nnz
    bits = kernels.HipKernels.gat_block_bits(h3).numpy().view(np.uint32)        # [block][w][lane][word]
    assert bits.shape == (h3.vals3.shape[0], 8, 64, 4)
    assert int(sum(bin(int(x)).count("1") for x in bits.reshape(-1))) == h3.nnz
    rr, cc, _ = h3.coo
    # every stored entry sets exactly its bit: block of the entry from the origins
    r0, c0 = h3.blk_row0.to(torch.int64), h3.blk_col0.to(torch.int64)
    key = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(r0, c0))}
    rng = np.random.default_rng(0)
    for e in rng.choice(rr.numel(), 400, replace=False):
        i, j = int(rr[e]), int(cc[e])
        k = next(kk for (a, b), kk in key.items() if a <= i < a + 512 and b <= j < b + 128)
        il, kl = i - int(r0[k]), j - int(c0[k])
        w, rb, lo = il // 64, (il // 32) % 2, il % 32
        ks, hk, e8 = kl // 16, (kl // 8) % 2, kl % 8
        u = 2 * ks + rb
        word = int(bits[k, w, 32 * hk + lo, u >> 2])
        assert (word >> (8 * (u & 3) + e8)) & 1, (i, j)
        # ... and it is where dense3_index puts the value
        assert float(h3.vals3[k, partition.dense3_index(il, kl)]) == 1.0


gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    d = torch.device("cuda:0")
    torch.cuda.set_device(d)
    return d


@pytest.fixture(scope="module")
def K(dev):
    return pkg("kernels").HipKernels(dev)


def _structures(K, A, tau):
    """(whole pattern, remaining entries, blocks) of A on the device."""
    partition, gat = pkg("partition"), pkg("gat")
    nr, nc = A.shape
    r, c = _coords(A)
    ones = torch.ones(r.numel())
    full = partition.csr_from_coo(r, c, ones, nr, nc, nslices=1, core=False)
    keep, h3 = partition.split_dense3(r, c, ones, nr, nc, tau)
    assert h3 is not None
    rest = partition.csr_from_coo(r[keep], c[keep], ones[:int(keep.sum())], nr, nc, nslices=1, core=False)
    return (K.prepare_gat(full, *gat._row_lists(full.rowptr, 1024)), K.prepare_gat(rest, *gat._row_lists(rest.rowptr, 1024)),
            K.prepare_gat_blocks(h3), h3)


@gpu
@pytest.mark.parametrize("n,m,heads", [(1100, 700, 4), (512, 128, 4), (1500, 900, 3), (600, 1030, 1), (2100, 400, 2)])
def test_forward_blocks_vs_gather_kernel_and_oracle(K, dev, n, m, heads):
    d, F = 64, heads * 64
    A, rng = _corner_graph(n, m, n + heads)
    dA, dR, G, h3 = _structures(K, A, 0.06)
    assert G.nnz + dR.nnz == A.nnz and G.nnz > 0.3 * A.nnz
    pw2 = F + (heads + 3) // 4 * 4
    ld = F + heads + (4 - heads % 4) % 4 + 4
    Zc = (rng.standard_normal((m, ld)) * 0.7).astype(np.float32)
    s1 = (rng.standard_normal((n, heads)) * 1.5).astype(np.float32)
    s2 = np.ascontiguousarray(Zc[:, F:F + heads])
    Zd, s1d, s2d = torch.from_numpy(Zc).to(dev), torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)
    rowstat = torch.full((n, heads, 4), float("nan"), device=dev)
    beta = torch.zeros((n, heads), device=dev)
    K.gat_edge_softmax(dA, s1d, s2d, heads, 0.2, 0, m, None, beta, rowstat)
    ref, refV = torch.full((n, F), float("nan"), device=dev), torch.full((n, pw2), float("nan"), device=dev)
    assert K.spmm_heads_forward2(dA, rowstat, s2d, 0.2, 0, Zd, ref, refV, heads, d)
    out, V = torch.full((n, F), float("nan"), device=dev), torch.full((n, pw2), float("nan"), device=dev)
    assert K.spmm_heads_forward2(dR, rowstat, s2d, 0.2, 0, Zd, out, V, heads, d)
    assert K.gat_blocks_forward(G, rowstat, s2d, 0.2, Zd, out, V, heads, d)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.isfinite(V).all()
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) < 4e-6 * scale
    assert float((V - refV).abs().max()) < 4e-6 * float(refV.abs().max())
    exp = oracle.gat_aggregate_np(A, Zc[:, :F].astype(np.float64), s1.astype(np.float64), s2.astype(np.float64), "standard", 0.2, m,
                                  Zc[:, :F].sum(0).astype(np.float64))
    assert rel_err(out.cpu().numpy(), exp) < 2e-5
    # the weights of a row sum to one: C = sum c_ij lies between slope and 1
    Ccol = V[:, F:F + heads].cpu().numpy()
    has = np.diff(A.indptr) > 0
    assert (Ccol[has] > 0.2 - 1e-5).all() and (Ccol[has] < 1 + 1e-5).all()
    # bit-reproducible (fixed slot order)
    out2, V2 = torch.empty_like(out), torch.empty_like(V)
    K.spmm_heads_forward2(dR, rowstat, s2d, 0.2, 0, Zd, out2, V2, heads, d)
    K.gat_blocks_forward(G, rowstat, s2d, 0.2, Zd, out2, V2, heads, d)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(V, V2)


@gpu
@pytest.mark.parametrize("n,m,heads", [(1100, 700, 4), (512, 128, 4), (1500, 900, 3), (600, 1030, 1), (2100, 400, 2)])
def test_backward_blocks_vs_gather_kernel_and_oracle(K, dev, n, m, heads):
    """The transposed pattern: rows j (m of them), columns i (n)."""
    d, F = 64, heads * 64
    A, rng = _corner_graph(n, m, n + heads)
    dA, _, _, _ = _structures(K, A, 0.06)
    AT = sp.csr_matrix(A.T)
    AT.sort_indices()
    dT, dTR, GT, _ = _structures(K, AT, 0.06)
    pw = F + (heads + 3) // 4 * 4
    ld = pw + 4
    Zc = (rng.standard_normal((m, ld)) * 0.7).astype(np.float32)
    s1 = (rng.standard_normal((n, heads)) * 1.5).astype(np.float32)
    s2 = np.ascontiguousarray(Zc[:, F:F + heads])
    dOut = rng.standard_normal((n, F)).astype(np.float32)
    Zd, s1d, s2d, dOd = (torch.from_numpy(x).to(dev) for x in (Zc, s1, s2, dOut))
    rowstat = torch.full((n, heads, 4), float("nan"), device=dev)
    K.gat_edge_softmax(dA, s1d, s2d, heads, 0.2, 0, m, None, torch.zeros((n, heads), device=dev), rowstat)
    out, V = torch.empty((n, F), device=dev), torch.empty((n, pw), device=dev)
    assert K.spmm_heads_forward2(dA, rowstat, s2d, 0.2, 0, Zd, out, V, heads, d)
    t = (dOd.view(n, heads, d) * out.view(n, heads, d)).sum(-1).contiguous()
    ref = torch.full((m, ld), float("nan"), device=dev)
    assert K.spmm_heads_grad(dT, rowstat, s2d, 0.2, 0, dOd, Zd, t, ref, None, heads, d)
    got = torch.full((m, ld), float("nan"), device=dev)
    assert K.spmm_heads_grad(dTR, rowstat, s2d, 0.2, 0, dOd, Zd, t, got, None, heads, d)
    assert K.gat_blocks_backward(GT, rowstat, s2d, 0.2, dOd, Zd, t, got, heads, d)
    torch.cuda.synchronize()
    assert torch.isnan(got[:, pw:]).all()                                  # columns beyond the output row untouched
    g, r = got[:, :F + heads], ref[:, :F + heads]
    assert torch.isfinite(g).all()
    assert float((g[:, :F] - r[:, :F]).abs().max()) < 4e-6 * float(r[:, :F].abs().max())
    # ds2 is a difference of two sums of the size of sum |c| |dp|: held to that scale
    assert float((g[:, F:] - r[:, F:]).abs().max()) < 2e-5 * max(float(r[:, F:].abs().max()), 1.0)
    edZ, _, eds2 = oracle.gat_aggregate_backward_np(A, Zc[:, :F].astype(np.float64), s1.astype(np.float64), s2.astype(np.float64),
                                                    dOut.astype(np.float64), "standard", 0.2, m, Zc[:, :F].sum(0).astype(np.float64), None)
    assert rel_err(g[:, :F].cpu().numpy(), edZ) < 2e-5
    assert rel_err(g[:, F:].cpu().numpy(), eds2) < 1e-4


@gpu
def test_blocks_refuse_what_they_do_not_cover(K, dev):
    A, rng = _corner_graph(1100, 700, 1)
    dA, dR, G, _ = _structures(K, A, 0.06)
    n, m, heads, d = 1100, 700, 2, 32
    F = heads * d
    z = torch.zeros((m, F + 4), device=dev)
    assert not K.gat_blocks_forward(G, torch.zeros((n, heads, 4), device=dev), z[:, F:F + heads].contiguous(), 0.2, z,
                                    torch.zeros((n, F), device=dev), torch.zeros((n, F + 4), device=dev), heads, d)
    L = pkg("_lib").lib()
    assert L.pgcn_gat_blocks_forward_f32(None, 1, None, None, None, None, 1, None, None, 4, 0.2, 4, 64, 10, 10, None, 256, None, 0, None, 0, 0,
                                         None) == -1
    assert b"pgcn_gat_blocks_forward_f32" in L.pgcn_last_error()
    assert L.pgcn_gat_blocks_forward_f32(None, 1, None, None, None, None, 1, None, None, 4, 0.2, 4, 32, 10, 10, None, 256, None, 0, None, 0, 0,
                                         None) == pkg("_lib").PGCN_EUNSUPPORTED


@gpu
@pytest.mark.parametrize("shape", ["corner", "communities"])
def test_engine_with_and_without_blocks(dev, shape):
    """One rank, forward / backward of the aggregation: the engine built with the blocks against the engine built without them (the
    path test_gat_gpu.py holds to the reference's layers).  "communities": two planted communities, so that the vertex order has bands and
    the block grid restarts inside the matrix (blocks of any origin, bands that end inside a block)."""
    import dataclasses
    partition, gat, kernels, tuning = pkg("partition"), pkg("gat"), pkg("kernels"), pkg("tuning")
    if shape == "corner":
        n = 1300
        A, rng = _corner_graph(n, n, 11, fill=0.25, rows=800, cols=500)
    else:
        n = 5000                                             # ten planted communities of 470-530 vertices: the order becomes a community order
        rng = np.random.default_rng(12)
        lab = np.minimum(np.arange(n) // 500 + (rng.random(n) < 0.06), 9)
        same = lab[:, None] == lab[None, :]
        A = sp.csr_matrix(((rng.random((n, n)) < np.where(same, 0.2, 0.0005))).astype(np.float32))
    A = sp.csr_matrix(((A + A.T) > 0).astype(np.float32))
    A.setdiag(1)
    Ac = sp.coo_matrix(A)
    row, col = torch.from_numpy(Ac.row.astype(np.int64)), torch.from_numpy(Ac.col.astype(np.int64))
    band_min = partition.ORDER_BAND_MIN
    partition.ORDER_BAND_MIN = 400                           # (a band per community at this size)
    try:
        part = partition.build_partition(row, col, torch.ones(row.numel()), n, torch.zeros(n, dtype=torch.int64), 0, 1)
    finally:
        partition.ORDER_BAND_MIN = band_min
    Kp = kernels.HipKernels(dev)
    saved = gat._T
    heads, d = 4, 64
    F = heads * d
    res = {}
    try:
        for on in (True, False):
            gat._T = dataclasses.replace(saved, gat_blocks=on)
            eng = gat.GatEngine(part, Kp, dev, None)
            assert (eng.fwd_blocks is not None) == on
            if on:
                assert eng.blocks_nnz > 0.2 * eng.nnz
                if shape == "communities":
                    assert part.local_bands is not None and part.local_bands.numel() >= 2
                    assert int((eng.fwd_blocks.work_row0 % 512 != 0).sum()) > 0      # a block row that starts at a band, not on the global grid
            g = torch.Generator().manual_seed(5)
            Z = (torch.randn(n, F, generator=g) * 0.7).to(dev)
            s1, s2 = (torch.randn(n, heads, generator=g) * 1.5).to(dev), (torch.randn(n, heads, generator=g) * 1.5).to(dev)
            dOut = torch.randn(n, F, generator=g).to(dev)
            st = eng.new_layer_state(heads, d)
            out = eng.forward(st, Z, s1, s2)
            dZ, ds1, ds2 = eng.backward(st, dOut)
            torch.cuda.synchronize()
            res[on] = [x.clone() for x in (out, dZ, ds1, ds2)]
    finally:
        gat._T = saved
    for a, b, tol in zip(res[True], res[False], (4e-6, 4e-6, 2e-5, 2e-5)):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) < tol * max(float(b.abs().max()), 1.0)
