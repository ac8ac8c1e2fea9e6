"""Host side of the bf16 three-plane MFMA blocks (pgcn_spmm_dense_bf16x3_f32, csrc/pgcn_spmm_dense3.hip): the exact
three-way split the kernels apply to both operands (partition.bf16_split3 restates split_pair), the A-operand block
layout (partition.dense3_index / HostDense3.vals3) and the arithmetic of the six-product scheme emulated in numpy
(every partial product exact in fp32, fp32 accumulation) against float64.  The kernel itself is held to the oracle by
tests/test_hip_gpu.py::test_spmm_bf16x3_blocks (its r04 micro-benchmark, tools/micro/dense3_bench.cpp, is in the git history)."""
import numpy as np
import scipy.sparse as sp
import torch

from conftest import pkg

planes_mod = pkg("partition")


def _is_bf16(x: np.ndarray) -> bool:
    return bool(((x.view(np.uint32) & 0xFFFF) == 0).all())


def test_split_is_exact_and_every_term_is_bf16():
    partition = planes_mod
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.standard_normal(20000).astype(np.float32),
        (rng.standard_normal(5000) * 1e-30).astype(np.float32),
        (rng.standard_normal(5000) * 1e30).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -3.0e38, 2.0 ** -126, 2.0 ** -130,
                  0.00390625, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -9], dtype=np.float32),       # ties, binade edges, denormals
        rng.integers(0, 2 ** 31 - 2 ** 24, 20000).astype(np.uint32).view(np.float32),           # any finite bit pattern
    ])
    x = x[np.isfinite(x)]
    x1, x2, x3 = (t.numpy() for t in partition.bf16_split3(torch.from_numpy(x)))
    assert _is_bf16(x1) and _is_bf16(x2) and _is_bf16(x3)
    total = x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64)
    big = (np.abs(x) >= 2.0 ** -110) | (x == 0)
    np.testing.assert_array_equal(total[big], x.astype(np.float64)[big])
    assert (~big).any() and (np.abs(total - x)[~big] <= 2.0 ** -134).all()       # under the smallest bf16 denormal: rounded
    # round to nearest: the first term is within half a bf16 ulp, the second within 2^-17 relative
    ok = np.abs(x) > 1e-30
    assert (np.abs(x - x1)[ok] <= np.abs(x[ok]) * 2.0 ** -8).all()
    assert (np.abs(x - x1 - x2)[ok] <= np.abs(x[ok]) * 2.0 ** -16).all()
    # ties go to even (bit 16 of the result is 0 for an exact tie)
    tie = np.array([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8], dtype=np.float32)
    t1 = partition.bf16_round(torch.from_numpy(tie)).numpy()
    np.testing.assert_array_equal(t1, np.array([1.0, 1.0 + 2.0 ** -6], dtype=np.float32))
    inf = torch.tensor([float("inf"), float("-inf"), float("nan")])
    r = partition.bf16_round(inf)
    assert torch.isinf(r[0]) and r[0] > 0 and torch.isinf(r[1]) and r[1] < 0 and torch.isnan(r[2])


def test_blocks_are_stored_in_the_a_operand_order_of_the_bf16_mfma():
    """vals3[b][w][unit = 2 ks + rb][h][lane][e] = A[64 w + 32 rb + (lane & 31)][16 ks + 8 (lane >> 5) + 4 h + e] of block b;
    blocks below the fill threshold and whatever the blocks do not hold stay in the other parts; the panel list is
    the sorted set of the blocks' column blocks."""
    partition = pkg("partition")
    rng = np.random.default_rng(3)
    n, m = 1100, 700
    D = ((rng.random((n, m)) < 0.3) * rng.standard_normal((n, m))).astype(np.float32)
    D[600:, 300:] *= rng.random((500, 400)) < 0.1
    h = partition.csr_from_scipy(sp.coo_matrix(D), nslices=1, core=True, strip=True, strip_min=32, dense3_tau=0.2)
    hd = h.dense3
    assert hd is not None and h.nnz == int((D != 0).sum())
    nb = hd.vals3.shape[0]
    assert tuple(hd.vals3.shape) == (nb, 512 * 128) and hd.vals3.dtype == torch.float32
    Dp = np.zeros((1536, 768), np.float32); Dp[:n, :m] = D
    i, k = np.meshgrid(np.arange(512), np.arange(128), indexing="ij")
    idx = partition.dense3_index(i, k)
    assert np.array_equal(np.sort(idx.ravel()), np.arange(512 * 128))                       # a permutation of the block
    w, rb, il, ks, hk, hh, e = i // 64, (i // 32) % 2, i % 32, k // 16, (k // 8) % 2, (k // 4) % 2, k % 4
    assert np.array_equal(idx, ((((w * 16 + 2 * ks + rb) * 2 + hh) * 64) + 32 * hk + il) * 4 + e)
    held = np.zeros_like(Dp, dtype=bool)
    for b in range(nb):
        br, bp = int(hd.blk_row[b]), int(hd.blk_panel[b])
        blk = Dp[br * 512:(br + 1) * 512, bp * 128:(bp + 1) * 128]
        np.testing.assert_array_equal(hd.vals3[b].numpy()[idx], blk)
        assert (blk != 0).sum() >= 0.2 * 512 * 128
        held[br * 512:(br + 1) * 512, bp * 128:(bp + 1) * 128] = True
        assert int(hd.panel_list[hd.blk_img[b]]) == 128 * bp and int(hd.blk_row0[b]) == 512 * br and int(hd.blk_col0[b]) == 128 * bp
    assert torch.equal(hd.panel_list, 128 * torch.unique(hd.blk_panel)) and nb >= 5
    r, c, v = hd.coo
    assert held[r.numpy(), c.numpy()].all() and r.numel() == int(((Dp != 0) & held).sum())
    # pieces: runs of blocks of ONE block row, every block in exactly one piece, 512 slot rows per piece
    seen = np.zeros(nb, np.int32)
    for br, first, cnt, slot0 in hd.work.numpy():
        assert (hd.blk_row[first:first + cnt].numpy() == br).all() and slot0 % 512 == 0
        seen[first:first + cnt] += 1
    assert (seen == 1).all() and hd.nslots == hd.npieces * 512


def test_six_products_reach_fp32_accuracy():
    """The kernel's arithmetic, emulated: a.h from the six partial products a1h1, a1h2, a2h1, a1h3, a2h2, a3h1
    (each exact in fp32), summed in fp32 over k -- against float64.  Error within 2^-21 sum |a||h| per output
    (the documented worst case), and in practice of the size of an fp32 dot product's own rounding."""
    partition = planes_mod
    rng = np.random.default_rng(5)
    K, N = 128, 64
    A = (rng.random((128, K)) < 0.3) * rng.random((128, K)).astype(np.float32) * 0.1
    A = A.astype(np.float32)
    H = (rng.standard_normal((K, N)) * np.exp(rng.standard_normal((K, 1)) * 3)).astype(np.float32)   # rows of very different scale
    a = [t.numpy() for t in partition.bf16_split3(torch.from_numpy(A))]
    h = [t.numpy() for t in partition.bf16_split3(torch.from_numpy(H))]
    acc = np.zeros((128, N), np.float32)
    for k0 in range(0, K, 16):                       # one MFMA = 16 k values; products are exact, sums fp32
        for (p, q) in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):
            prod = a[p][:, k0:k0 + 16, None] * h[q][None, k0:k0 + 16, :]
            assert np.array_equal(prod.astype(np.float64), a[p][:, k0:k0 + 16, None].astype(np.float64) * h[q][None, k0:k0 + 16, :])
            acc = (acc + prod.sum(1, dtype=np.float32)).astype(np.float32)
    ref = A.astype(np.float64) @ H.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(H).astype(np.float64)
    err = np.abs(acc - ref)
    assert (err <= 2.0 ** -21 * scale + 1e-300).all()
    fp32 = (A @ H).astype(np.float64)
    assert err.max() <= 4 * max(np.abs(fp32 - ref).max(), 1e-30) + 2.0 ** -23 * scale.max()
