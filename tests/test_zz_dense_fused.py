"""relu(x . W^T) and its input gradient as the package's own bf16-split MFMA kernels (gemm/pgcn_dense.hip, include/pgcn_gemm.h,
PGCN.linear_relu_fused / linear_relu_grad_input_fused, tuning.dense_fused) -- replaces `F.relu(self.linear(AH))` of
/root/reference/GPU/PGCN.py:146-147 and the autograd of those two lines.

CPU: the library exports the entry points; a HOST build of the kernel's own index functions (gemm/pgcn_dense_tile.h -- LDS
image slots, operand lanes, accumulator layout, the unpredicated path of inner tiles and the guarded one of ragged tiles --
compiled by tests/native/pgcn_dense_emu.cpp and run lane by lane around an emulated v_mfma_f32_32x32x16_bf16) reproduces the
products to the error class of an fp32 dot product on full, ragged and tiny shapes; the ctypes binding and the autograd node
are driven through that build.  GPU: the same comparisons on the real kernels."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

PKG_DIR = os.path.join(ROOT, "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd")
SRC = os.path.join(ROOT, "tests", "native", "pgcn_dense_emu.cpp")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# bound on |result - float64| / sum |a||b|: six of the nine partial products, fp32 accumulation (observed 1.4e-7 .. 4e-7)
BOUND = 1e-6


def test_library_exports_the_dense_entry_points():
    P = pkg("PGCN")
    src = open(os.path.join(ROOT, "include", "pgcn_gemm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(pgcn_(?:linear|dense|wgrad|sign)_[a-z0-9_]+)\s*\(", src)))
    assert names == ["pgcn_dense_last_error", "pgcn_linear_relu_f32", "pgcn_linear_relu_grad_input_f32", "pgcn_linear_weight_grad_f32",
                     "pgcn_linear_weight_grad_ws_elems", "pgcn_sign_mask_f32", "pgcn_wgrad_last_error"]
    L = ctypes.CDLL(P.GEMM_LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libpgcn_gemm.so does not export %s" % n
    assert P.bind_dense_library(P.GEMM_LIB_PATH).pgcn_linear_relu_f32.argtypes is not None


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    """tests/native/pgcn_dense_emu.cpp (the kernel's index arithmetic on the host), bound like the library."""
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ for the host build of gemm/pgcn_dense_tile.h")
    out = str(tmp_path_factory.mktemp("dense_emu") / "libpgcn_dense_emu.so")
    subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-pass-failed",
                           "-I", os.path.join(PKG_DIR, "gemm"), SRC, "-o", out])
    return pkg("PGCN").bind_dense_library(out)


def _rel(got, want, den):
    return float(((got.double() - want).abs() / (den + 1e-30)).max())


def _pack_mask(y):
    """The sign mask of y as the kernels lay it out (include/pgcn_gemm.h): int32 [n, ceil(N / 32)], bit b of word w = (y[:, 32 w + b] > 0)."""
    n, N = y.shape
    mw = (N + 31) // 32
    bits = torch.zeros((n, mw * 32), dtype=torch.int64, device=y.device)
    bits[:, :N] = (y > 0).to(torch.int64)
    words = (bits.view(n, mw, 32) << torch.arange(32, device=y.device)).sum(-1)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)


def _check_pair(P, L, n, fin, fout, stream, dev="cpu", pad=0, seed=0):
    """forward (+ sign mask) + input gradient of one shape through the entry points of L against float64; returns the two errors."""
    g0 = torch.Generator().manual_seed(seed)
    x = torch.randn(n, fin + pad, generator=g0)[:, :fin].to(dev) if pad else torch.randn(n, fin, generator=g0).to(dev)
    w = (torch.randn(fout, fin, generator=g0) / 8).to(dev)
    g = torch.randn(n, fout, generator=g0).to(dev)
    y = P.linear_relu_call(L, x, w, True, stream)
    assert y is not None and y.shape == (n, fout)
    want = (x.double() @ w.double().t()).clamp_min(0)
    den = x.double().abs() @ w.double().abs().t()
    ef = _rel(y, want, den) if n else 0.0
    ylin = P.linear_relu_call(L, x, w, False, stream)
    assert torch.equal(ylin.clamp_min(0), y)                          # relu = 0: the same product, unclamped
    y2, mask = P.linear_relu_call(L, x, w, True, stream, want_mask=True)
    assert torch.equal(y2, y) and mask.shape == (n, P.mask_words(fout)) and mask.dtype is torch.int32
    assert torch.equal(mask, _pack_mask(y))                           # 1 bit per element, exactly (y > 0)
    assert torch.equal(P.sign_mask_call(L, y, stream), mask)          # ... and the stand-alone mask of an existing matrix
    gm, gx = P.linear_relu_grad_input_call(L, g, mask, w, stream)
    assert torch.equal(gm, torch.where(y > 0, g, torch.zeros((), device=dev)))      # threshold_backward, exactly
    wantx = gm.double() @ w.double()
    eb = _rel(gx, wantx, gm.double().abs() @ w.double().abs()) if n else 0.0
    gm0, gx0 = P.linear_relu_grad_input_call(L, g, None, w, stream)   # no mask: the plain product
    assert torch.equal(gm0, g)
    if n:
        assert _rel(gx0, g.double() @ w.double(), g.double().abs() @ w.double().abs()) <= BOUND
    return ef, eb


SHAPES = [(77, 128, 128), (32, 64, 64), (100, 36, 128), (65, 128, 40), (5, 8, 4), (33, 4, 4), (1, 128, 128), (64, 16, 100), (0, 128, 128)]


@pytest.mark.parametrize("n,fin,fout", SHAPES)
def test_host_build_reproduces_the_products(emu, n, fin, fout):
    ef, eb = _check_pair(pkg("PGCN"), emu, n, fin, fout, None)
    assert ef <= BOUND and eb <= BOUND, (ef, eb)


@pytest.mark.parametrize("n,fin,fout", [(96, 128, 128), (64, 64, 64), (97, 64, 128), (200, 128, 64), (50, 128, 44), (10, 8, 8)])
def test_host_build_inner_and_ragged_tiles(emu, n, fin, fout):
    """Tiles inside a full-width operand take the unpredicated loads / stores, a wave's last tile and ragged widths the guarded
    ones; both within the bound, and rows computed on either path agree bit for bit (a row's result does not depend on n)."""
    P = pkg("PGCN")
    ef, eb = _check_pair(P, emu, n, fin, fout, None)
    assert ef <= BOUND and eb <= BOUND, (ef, eb)
    g0 = torch.Generator().manual_seed(9)
    x, w = torch.randn(n, fin, generator=g0), torch.randn(fout, fin, generator=g0)
    y = P.linear_relu_call(emu, x, w, True, None)
    m = n - 3                                       # the same rows as part of a shorter matrix: other tiles become ragged
    assert torch.equal(P.linear_relu_call(emu, x[:m], w, True, None), y[:m])


def test_host_build_padded_rows_and_in_place_mask(emu):
    P = pkg("PGCN")
    ef, eb = _check_pair(P, emu, 70, 64, 64, None, pad=8)            # leading dimension 72 > width 64
    assert ef <= BOUND and eb <= BOUND
    # Gm == G is allowed by the header: every 16-byte piece is read and written by the one lane that owns it
    torch.manual_seed(3)
    g, y, w = torch.randn(50, 32), torch.randn(50, 32), torch.randn(32, 16)
    want = torch.where(y > 0, g, torch.zeros(()))
    mask = _pack_mask(y)
    gx = torch.empty(50, 16)
    rc = emu.pgcn_linear_relu_grad_input_f32(g.data_ptr(), 32, mask.data_ptr(), g.data_ptr(), 32, 50, 32, w.data_ptr(), 16, 16,
                                             gx.data_ptr(), 16, None)
    assert rc == 0 and torch.equal(g, want) and torch.allclose(gx, want @ w, atol=1e-4)
    # no Gm asked for: only the product
    gx2 = torch.empty(50, 16)
    rc = emu.pgcn_linear_relu_grad_input_f32(want.data_ptr(), 32, mask.data_ptr(), None, 0, 50, 32, w.data_ptr(), 16, 16,
                                             gx2.data_ptr(), 16, None)
    assert rc == 0 and torch.equal(gx2, gx)


def test_refusals_are_minus_two_and_errors_minus_one(emu):
    P = pkg("PGCN")
    x = torch.randn(10, 132)
    assert P.linear_relu_call(emu, x, torch.randn(8, 132), True, None) is None          # wider than 128
    assert P.linear_relu_call(emu, torch.randn(10, 6), torch.randn(8, 6), True, None) is None   # rows are not 16-byte pieces
    assert P.linear_relu_call(emu, torch.randn(10, 17)[:, 1:], torch.randn(8, 16), True, None) is None  # misaligned base / ld
    assert P.linear_relu_call(emu, torch.randn(10, 8), torch.randn(8, 4), True, None) is None    # widths disagree: not called
    assert P.linear_relu_grad_input_call(emu, torch.randn(10, 6), None, torch.randn(6, 8), None) is None
    assert P.linear_relu_grad_input_call(emu, torch.randn(10, 8), torch.zeros(10, 2, dtype=torch.int32), torch.randn(8, 8), None) is None   # mask of another width
    y = torch.empty(4, 4)
    assert emu.pgcn_linear_relu_f32(None, 4, 4, 4, torch.randn(4, 4).data_ptr(), 4, 4, y.data_ptr(), 4, 1, None, None) == -1
    assert b"bad argument" in emu.pgcn_dense_last_error()
    assert emu.pgcn_linear_relu_f32(y.data_ptr(), 2, 4, 4, y.data_ptr(), 4, 4, y.data_ptr(), 4, 1, None, None) != 0   # ld below the width


def test_autograd_node_through_the_host_build(emu, monkeypatch):
    """PGCN._LinearReluNoBias with tuning.dense_fused >= 2 takes both entry points (here: the host build on CPU tensors), the
    sign mask travelling from the forward to the backward instead of y, and agrees with the stock route; level 0 never touches them."""
    P, tuning = pkg("PGCN"), pkg("tuning")
    calls = []
    monkeypatch.setattr(P, "_dense_operand_ok", lambda *ts: all(t.dim() == 2 and t.stride(1) == 1 for t in ts))
    monkeypatch.setattr(P, "_dense_stream", lambda t: None)
    monkeypatch.setattr(P, "_dense_lib", lambda: (calls.append(1), emu)[1])
    torch.manual_seed(1)
    x0, w0 = torch.randn(90, 64), torch.randn(32, 64) / 8
    out = {}
    for level in (0, 1, 2, 3):
        monkeypatch.setattr(tuning.T, "dense_fused", level)
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        calls.clear()
        y = P._LinearReluNoBias.apply(x, w)
        (y * torch.arange(32.0)).sum().backward()
        out[level] = (y.detach(), x.grad, w.grad, len(calls))
    assert [out[l][3] for l in (0, 1, 2, 3)] == [0, 1, 2, 3]       # (level 3 asks the library for the weight gradient too: the host build has none)
    for level in (1, 2, 3):
        for a, b in zip(out[level][:3], out[0][:3]):                    # (sums with cancellation: relative to the tensor's scale)
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (level, float((a - b).abs().max()))
    assert torch.equal(out[1][0], out[2][0])
    # operands the kernel refuses (width 132) fall through to the library product under any level
    monkeypatch.setattr(tuning.T, "dense_fused", 2)
    xw, ww = torch.randn(20, 132, requires_grad=True), torch.randn(8, 132, requires_grad=True)
    yw = P._LinearReluNoBias.apply(xw, ww)
    yw.sum().backward()
    assert torch.allclose(yw, (xw @ ww.t()).clamp_min(0), atol=1e-5) and xw.grad is not None


def test_cpu_tensors_never_reach_the_kernels():
    P, tuning = pkg("PGCN"), pkg("tuning")
    assert tuning.Tuning().dense_fused == 3 and not hasattr(tuning.Tuning(), "layer_fused")   # (tuning.py / HISTORY.md have the epochs)
    assert P.linear_relu_fused(torch.randn(8, 8), torch.randn(4, 8)) is None
    assert P.linear_relu_grad_input_fused(torch.randn(8, 4), None, torch.randn(4, 8)) is None
    assert P.weight_grad_fused(torch.randn(8, 4), torch.randn(8, 8)) is None
    y = torch.randn(5, 70)
    assert torch.equal(P.unpack_sign_mask(_pack_mask(y), 70), y > 0)


# ---- GPU ------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("n,fin,fout", SHAPES + [(232965, 128, 128), (100003, 64, 64), (4097, 128, 44)])
def test_kernels_reproduce_the_products(n, fin, fout):
    P = pkg("PGCN")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ef, eb = _check_pair(P, P._dense_lib(), n, fin, fout, torch.cuda.current_stream(dev).cuda_stream, dev=dev)
    torch.cuda.synchronize()
    assert ef <= BOUND and eb <= BOUND, (ef, eb)


@pytest.mark.gpu
def test_kernels_against_the_library_route_and_reproducible():
    """At the benchmark layer shape: no further from float64 than twice the rocBLAS / PyTorch product, bit-identical run to run,
    on a side stream as well, padded rows taken."""
    P = pkg("PGCN")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n, f = 232965, 128
    g0 = torch.Generator().manual_seed(7)
    x = torch.randn(n, f, generator=g0).to(dev)
    w = (torch.randn(f, f, generator=g0) / 11).to(dev)
    y = P.linear_relu_fused(x, w)
    assert y is not None, "the kernel refused the benchmark shape"
    stock = (x @ w.t()).clamp_min_(0)
    exact = (x.double() @ w.double().t()).clamp_min_(0)
    e_mine, e_stock = float((y.double() - exact).abs().max()), float((stock.double() - exact).abs().max())
    assert e_mine <= max(2 * e_stock, 1e-6 * float(exact.abs().max())), (e_mine, e_stock)
    assert torch.equal(P.linear_relu_fused(x, w), y)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        y2 = P.linear_relu_fused(x, w)
    side.synchronize()
    assert torch.equal(y2, y)
    g = torch.randn(n, f, generator=g0).to(dev)
    ym, mask = P.linear_relu_fused(x, w, want_mask=True)
    assert torch.equal(ym, y) and torch.equal(mask, _pack_mask(y))
    gm, gx = P.linear_relu_grad_input_fused(g, mask, w)
    assert torch.equal(gm, torch.ops.aten.threshold_backward(g, y, 0.0))
    exact = gm.double() @ w.double()
    e_mine, e_stock = float((gx.double() - exact).abs().max()), float(((gm @ w).double() - exact).abs().max())
    assert e_mine <= max(2 * e_stock, 1e-6 * float(exact.abs().max())), (e_mine, e_stock)
    assert torch.equal(P.linear_relu_grad_input_fused(g, mask, w)[1], gx)
    # the weight gradient at the same shape: against float64, no further than twice the library's, bit-identical run to run
    gw = P.weight_grad_fused(gm, x)
    assert gw is not None and gw.shape == (f, f)
    exact = gm.double().t() @ x.double()
    e_mine, e_stock = float((gw.double() - exact).abs().max()), float(((gm.t() @ x).double() - exact).abs().max())
    assert e_mine <= max(2 * e_stock, 1e-6 * float(exact.abs().max())), (e_mine, e_stock)
    assert torch.equal(P.weight_grad_fused(gm, x), gw)
    xp = torch.randn(5000, 2 * f, device=dev)[:, :f]                   # leading dimension 256
    assert torch.allclose(P.linear_relu_fused(xp, w), (xp @ w.t()).clamp_min_(0), atol=1e-4)
    assert P.linear_relu_fused(torch.randn(64, 132, device=dev), torch.randn(8, 132, device=dev)) is None


@pytest.mark.gpu
def test_layer_with_the_kernels_switched_on(monkeypatch):
    """The autograd node of PGCN.py:146-147 with tuning.dense_fused = 3 (all three products) against level 0 on the GPU."""
    P, tuning = pkg("PGCN"), pkg("tuning")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(2)
    x0, w0 = torch.randn(30011, 128, device=dev), torch.randn(128, 128, device=dev) / 11
    coef = torch.randn(30011, 128, device=dev)
    out = {}
    for level in (0, 3, 3):
        monkeypatch.setattr(tuning.T, "dense_fused", level)
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        y = P._LinearReluNoBias.apply(x, w)
        (y * coef).sum().backward()
        torch.cuda.synchronize()
        if level in out:            # reproducible: all three products are the package's own, fixed-order kernels
            assert torch.equal(out[level][0], y.detach()) and torch.equal(out[level][1], x.grad) and torch.equal(out[level][2], w.grad)
        out[level] = (y.detach(), x.grad, w.grad)
    for a, b in zip(out[3], out[0]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale


WG_SHAPES = [(232965, 128, 128), (100003, 64, 64), (4097, 128, 44), (77, 128, 128), (1000, 40, 128), (33, 4, 4), (15, 64, 128), (1, 128, 128),
             (5000, 100, 36), (16, 32, 32), (600, 96, 64), (0, 128, 128)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,fout,fin", WG_SHAPES)
def test_weight_gradient_kernel(n, fout, fin):
    """gm^T . x (gemm/pgcn_wgrad.hip) against float64: full, ragged and tiny shapes, rows that are not a multiple of a step, padded
    leading dimensions; bound 1e-6 of sum |g||x| per element."""
    P = pkg("PGCN")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g0 = torch.Generator().manual_seed(n + fout)
    gm = torch.randn(n, fout + 4, generator=g0).to(dev)[:, :fout]           # leading dimension fout + 4
    gm = gm * (torch.rand(n, fout, generator=g0).to(dev) > 0.4)             # (a masked gradient: zeros in it)
    x = torch.randn(n, fin, generator=g0).to(dev)
    dw = P.weight_grad_fused(gm, x)
    assert dw is not None and dw.shape == (fout, fin)
    want = gm.double().t() @ x.double()
    den = gm.double().abs().t() @ x.double().abs()
    err = float(((dw.double() - want).abs() / (den + 1e-30)).max()) if n else float(dw.abs().max())
    assert err <= BOUND, err
    assert torch.equal(P.weight_grad_fused(gm, x), dw)                      # fixed-order sums: bit-identical run to run


@pytest.mark.gpu
def test_weight_gradient_on_a_side_stream_and_refusals():
    P = pkg("PGCN")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(4)
    gm, x = torch.randn(20000, 128, device=dev), torch.randn(20000, 128, device=dev)
    ref = P.weight_grad_fused(gm, x)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        dw = P.weight_grad_fused(gm, x)
    side.synchronize()
    assert torch.equal(dw, ref)
    assert P.weight_grad_fused(torch.randn(64, 132, device=dev), torch.randn(64, 8, device=dev)) is None      # wider than 128
    assert P.weight_grad_fused(torch.randn(64, 8, device=dev), torch.randn(32, 8, device=dev)) is None        # rows disagree
