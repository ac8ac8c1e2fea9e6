"""Host side of the bf16 three-plane MFMA tiles (tools/experiments/dense3, prepared and harness-measured in r03,
not in the library yet): the exact three-way split, the A-operand plane layout built from the product's own MFMA
tile image, the arithmetic of the six-product scheme emulated in numpy (every partial product exact in fp32, fp32
accumulation) against float64, and that integrate.patch -- the wiring into partition.py / kernels.py / the C ABI --
still applies to the tree.  The kernel itself ran on the MI355X through tools/experiments/dense3/dense3_bench.cpp
(profiles/r03_dense3_bench.txt), whose C++ layout builder follows the same formula as the test below."""
import importlib.util
import os
import subprocess

import numpy as np
import scipy.sparse as sp
import torch

from conftest import pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "experiments", "dense3")
_spec = importlib.util.spec_from_file_location("dense3_planes", os.path.join(EXP, "planes.py"))
planes_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(planes_mod)


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


def test_planes_are_the_a_operand_order_of_the_bf16_mfma():
    """planes[t][w][ks][p][lane][j] = bf16 plane p of A[32 w + (lane & 31)][16 ks + 8 (lane >> 5) + j]."""
    partition = pkg("partition")
    rng = np.random.default_rng(3)
    n, m = 300, 384
    D = ((rng.random((n, m)) < 0.5) * rng.standard_normal((n, m))).astype(np.float32)
    h = partition.csr_from_scipy(sp.coo_matrix(D), nslices=1, core=True, tau=0.05, dense_tau=0.2, strip=False)
    hd = h.dense
    assert hd is not None
    nt = hd.tile_row.numel()
    planes = planes_mod.dense_planes(hd.vals)                                    # from the product's own fp32 tile image
    assert tuple(planes.shape) == (nt, 4, 8, 3, 64, 8) and planes.dtype == torch.int16
    pl = planes.numpy().view(np.uint16).astype(np.uint32) << 16
    pl = pl.view(np.float32)                                                     # the planes as fp32 numbers
    Dp = np.zeros((384, 384), np.float32); Dp[:n, :m] = D
    i, k = np.meshgrid(np.arange(128), np.arange(128), indexing="ij")
    w, il, ks, hk, j = i // 32, i % 32, k // 16, (k // 8) % 2, k % 8
    for t in range(nt):
        tr, tp = int(hd.tile_row[t]), int(hd.tile_panel[t])
        tile = Dp[tr * 128:(tr + 1) * 128, tp * 128:(tp + 1) * 128]
        terms = [pl[t][w, ks, p, 32 * hk + il, j] for p in range(3)]
        np.testing.assert_array_equal(terms[0].astype(np.float64) + terms[1] + terms[2], tile.astype(np.float64))
        x1, x2, x3 = (x.numpy() for x in planes_mod.bf16_split3(torch.from_numpy(tile.copy())))
        np.testing.assert_array_equal(terms[0], x1)
        np.testing.assert_array_equal(terms[1], x2)
        np.testing.assert_array_equal(terms[2], x3)


def test_integration_patch_applies():
    """integrate.patch = the edits that put the kernel into the library behind tuning.dense_bf16x3 (header entry,
    build.sh, ctypes signature, HostDense.planes, the dispatch in kernels.py): it must keep applying to the tree."""
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        import pytest
        pytest.skip("not a git checkout (a gpurun snapshot)")
    out = subprocess.run(["git", "apply", "--check", os.path.join(EXP, "integrate.patch")], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


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
