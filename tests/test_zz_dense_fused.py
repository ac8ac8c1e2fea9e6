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
    names = sorted(set(re.findall(r"\b(pgcn_(?:linear|dense|wgrad)_[a-z0-9_]+)\s*\(", src)))
    names += sorted(set(re.findall(r"\b(pgcn_fixup_[a-z0-9_]+)\s*\(", src)))
    assert names == ["pgcn_dense_last_error", "pgcn_linear_epilogue_f32", "pgcn_linear_relu_f32", "pgcn_linear_relu_grad_input_f32",
                     "pgcn_fixup_linear_f32"]
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


def _check_pair(P, L, n, fin, fout, stream, dev="cpu", pad=0, seed=0):
    """forward + input gradient of one shape through the entry points of L against float64; returns the two errors."""
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
    gm, gx = P.linear_relu_grad_input_call(L, g, y, w, stream)
    assert torch.equal(gm, torch.where(y > 0, g, torch.zeros((), device=dev)))      # threshold_backward, exactly
    wantx = gm.double() @ w.double()
    eb = _rel(gx, wantx, gm.double().abs() @ w.double().abs()) if n else 0.0
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
    gx = torch.empty(50, 16)
    rc = emu.pgcn_linear_relu_grad_input_f32(g.data_ptr(), 32, y.data_ptr(), 32, g.data_ptr(), 32, 50, 32, w.data_ptr(), 16, 16,
                                             gx.data_ptr(), 16, None)
    assert rc == 0 and torch.equal(g, want) and torch.allclose(gx, want @ w, atol=1e-4)
    # no Gm asked for: only the product
    gx2 = torch.empty(50, 16)
    rc = emu.pgcn_linear_relu_grad_input_f32(want.data_ptr(), 32, y.data_ptr(), 32, None, 0, 50, 32, w.data_ptr(), 16, 16,
                                             gx2.data_ptr(), 16, None)
    assert rc == 0 and torch.equal(gx2, gx)


def test_refusals_are_minus_two_and_errors_minus_one(emu):
    P = pkg("PGCN")
    x = torch.randn(10, 132)
    assert P.linear_relu_call(emu, x, torch.randn(8, 132), True, None) is None          # wider than 128
    assert P.linear_relu_call(emu, torch.randn(10, 6), torch.randn(8, 6), True, None) is None   # rows are not 16-byte pieces
    assert P.linear_relu_call(emu, torch.randn(10, 17)[:, 1:], torch.randn(8, 16), True, None) is None  # misaligned base / ld
    assert P.linear_relu_call(emu, torch.randn(10, 8), torch.randn(8, 4), True, None) is None    # widths disagree: not called
    assert P.linear_relu_grad_input_call(emu, torch.randn(10, 6), torch.randn(10, 6), torch.randn(6, 8), None) is None
    y = torch.empty(4, 4)
    assert emu.pgcn_linear_relu_f32(None, 4, 4, 4, torch.randn(4, 4).data_ptr(), 4, 4, y.data_ptr(), 4, 1, None) == -1
    assert b"bad argument" in emu.pgcn_dense_last_error()
    assert emu.pgcn_linear_relu_f32(y.data_ptr(), 2, 4, 4, y.data_ptr(), 4, 4, y.data_ptr(), 4, 1, None) != 0   # ld below the width


def test_autograd_node_through_the_host_build(emu, monkeypatch):
    """PGCN._LinearReluNoBias with tuning.dense_fused = 2 takes both entry points (here: the host build on CPU tensors) and
    agrees with the stock route; level 0 never touches them."""
    P, tuning = pkg("PGCN"), pkg("tuning")
    calls = []
    monkeypatch.setattr(P, "_dense_operand_ok", lambda *ts: all(t.dim() == 2 and t.stride(1) == 1 for t in ts))
    monkeypatch.setattr(P, "_dense_stream", lambda t: None)
    monkeypatch.setattr(P, "_dense_lib", lambda: (calls.append(1), emu)[1])
    torch.manual_seed(1)
    x0, w0 = torch.randn(90, 64), torch.randn(32, 64) / 8
    out = {}
    for level in (0, 1, 2):
        monkeypatch.setattr(tuning.T, "dense_fused", level)
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        calls.clear()
        y = P._LinearReluNoBias.apply(x, w)
        (y * torch.arange(32.0)).sum().backward()
        out[level] = (y.detach(), x.grad, w.grad, len(calls))
    assert [out[l][3] for l in (0, 1, 2)] == [0, 1, 2]
    for level in (1, 2):
        for a, b in zip(out[level][:3], out[0][:3]):                    # (sums with cancellation: relative to the tensor's scale)
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), (level, float((a - b).abs().max()))
    assert torch.equal(out[1][0], out[2][0])
    # operands the kernel refuses (width 132) fall through to the library product under any level
    monkeypatch.setattr(tuning.T, "dense_fused", 2)
    xw, ww = torch.randn(20, 132, requires_grad=True), torch.randn(8, 132, requires_grad=True)
    yw = P._LinearReluNoBias.apply(xw, ww)
    yw.sum().backward()
    assert torch.allclose(yw, (xw @ ww.t()).clamp_min(0), atol=1e-5) and xw.grad is not None


# ---- the fix-up of the aggregation as the loader of the product (pgcn_fixup_linear_f32) --------------------------------------------

def _random_deferred(n, f, seed, dev="cpu", max_slots=11, ids=True):
    """A random decomposition of an n x f matrix into partial rows: (row_fix, slot_ids, ws, base, S) with S the ordered fp32 sums
    exactly as csrc's spmm_fixup_list_kernel forms them (((0 + x0) + x1) + ...).  Some rows are `direct` (count -1: taken from
    base), some empty (count 0), some have more slots than one chunk of ids."""
    g0 = torch.Generator().manual_seed(seed)
    cnt = torch.randint(0, 6, (n,), generator=g0)
    cnt[torch.rand(n, generator=g0) < 0.1] = torch.randint(9, max_slots + 1, (1,), generator=g0).item()
    direct = torch.rand(n, generator=g0) < 0.15
    cnt[direct] = 0
    total = int(cnt.sum())
    nslots = total + 5
    ws = torch.randn(nslots, f, generator=g0)
    if ids:
        slot_ids = torch.randperm(nslots, generator=g0)[:total].to(torch.int32)
        begin = torch.cumsum(cnt, 0) - cnt
    else:                                             # consecutive slots: begin IS the first slot
        slot_ids, begin = None, torch.cumsum(cnt, 0) - cnt
    base = torch.randn(n, f, generator=g0)
    S = torch.zeros(n, f)
    for r in range(n):
        if direct[r]:
            S[r] = base[r]
            continue
        acc = torch.zeros(f)
        for t in range(int(cnt[r])):
            sid = int(slot_ids[begin[r] + t]) if ids else int(begin[r] + t)
            acc = acc + ws[sid]
        S[r] = acc
    row_fix = torch.stack([begin, torch.where(direct, torch.full_like(cnt, -1), cnt)], 1).to(torch.int32).contiguous()
    mv = lambda t: None if t is None else t.to(dev)
    return mv(row_fix), mv(slot_ids), mv(ws), mv(base), mv(S)


FIX_SHAPES = [(77, 128, 128), (64, 64, 64), (33, 16, 16), (100, 36, 128), (5, 8, 4), (1, 128, 128), (96, 128, 64), (200, 64, 128)]


def _check_fixup(P, L, n, fin, fout, stream, dev="cpu", ids=True):
    row_fix, slot_ids, ws, base, S = _random_deferred(n, fin, seed=n + fin, dev=dev, ids=ids)
    g0 = torch.Generator().manual_seed(5)
    w = (torch.randn(fout, fin, generator=g0) / 8).to(dev)
    # forward: relu(S . W^T), bit-identical to the separate route (the ordered sum, then the product kernel on it)
    y, none = P.fixup_linear_call(L, row_fix, slot_ids, ws, base, fin, w, True, P.EPI_RELU, None, False, stream)
    assert none is None and torch.equal(y, P.linear_relu_call(L, S, w, True, stream))
    # backward shape: S . W2 with the sum written out and the mask of the layer below folded in
    w2 = (torch.randn(fin, fout, generator=g0) / 8).to(dev)
    m = torch.randn(n, fout, generator=g0).to(dev)
    y2, S2 = P.fixup_linear_call(L, row_fix, slot_ids, ws, base, fin, w2, False, P.EPI_MASK, m, True, stream)
    assert torch.equal(S2, S)
    plain = P.linear_epilogue_call(L, S, w2, False, P.EPI_NONE, None, stream)
    assert torch.equal(y2, torch.where(m > 0, plain, torch.zeros((), device=dev)))
    assert torch.equal(P.linear_epilogue_call(L, S, w2, False, P.EPI_MASK, m, stream), y2)
    want = S.double() @ w2.double()
    den = S.double().abs() @ w2.double().abs()
    return _rel(plain, want, den) if n else 0.0


@pytest.mark.parametrize("n,fin,fout", FIX_SHAPES)
def test_host_build_of_the_fixup_loader(emu, n, fin, fout):
    """gemm/pgcn_dense_tile.h sum_half / store_half / store_c_masked around the emulated MFMA: slot lists with ids and consecutive,
    direct rows, empty rows, lists longer than one chunk of ids, ragged rows and widths."""
    P = pkg("PGCN")
    assert _check_fixup(P, emu, n, fin, fout, None) <= BOUND
    assert _check_fixup(P, emu, n, fin, fout, None, ids=False) <= BOUND


def test_fixup_loader_refusals(emu):
    P = pkg("PGCN")
    row_fix, slot_ids, ws, base, S = _random_deferred(10, 8, 1)
    assert P.fixup_linear_call(emu, row_fix, slot_ids, ws, base, 8, torch.randn(4, 6), True, 1, None, False, None) is None   # W does not fit
    rf6, ids6, ws6, base6, _ = _random_deferred(10, 6, 1)
    assert P.fixup_linear_call(emu, rf6, ids6, ws6, base6, 6, torch.randn(4, 6), True, 1, None, False, None) is None        # width % 4
    assert P.fixup_linear_call(emu, row_fix, slot_ids, ws, base, 8, torch.randn(8, 4), False, 2, None, False, None) is None  # mask missing
    y = torch.empty(10, 4)
    assert emu.pgcn_fixup_linear_f32(None, None, None, 8, base.data_ptr(), 8, 10, 8, torch.randn(4, 8).data_ptr(), 8, 4, 8, 1, None, 0,
                                     None, 0, y.data_ptr(), 4, 1, None) == -1


class _FakeEngine:
    """AggregationEngine stand-in on CPU tensors: A is dense, every aggregation comes back as a kernels.DeferredSum whose partial rows
    are a random split of the true rows (so that the loader's sum is exercised), as the HIP provider returns them."""

    def __init__(self, A):
        self.A = A
        self.calls = []

    def _split(self, X):
        K = pkg("kernels")
        n, f = X.shape
        g0 = torch.Generator().manual_seed(len(self.calls))
        a = torch.randn(n, f, generator=g0)
        ws = torch.cat([a, X - a])                        # two slots per row: a and (X - a)
        row_fix = torch.stack([2 * torch.arange(n), torch.full((n,), 2)], 1).to(torch.int32)
        slot_ids = torch.stack([torch.arange(n), n + torch.arange(n)], 1).reshape(-1).to(torch.int32)
        base = torch.empty(n, f)
        def finish():
            base.copy_(ws[:n] + ws[n:])
            return base
        return K.DeferredSum(row_fix, slot_ids, ws, base, f, finish)

    def forward_deferred(self, H):
        self.calls.append("f")
        return self._split(self.A @ H)

    def backward_deferred(self, G):
        self.calls.append("b")
        return self._split(self.A.t() @ G)


def test_fused_layer_node_through_the_host_build(emu, monkeypatch):
    """PGCN._AggLinearRelu (tuning.layer_fused): two stacked layers on the host build against plain autograd of the same two layers --
    the re-associated backward (T = A^T.Gm, dH = T.W, dW = T^T.H), the mask of the layer below folded into the upper layer's input
    gradient, and the lower layer recognising the pre-masked gradient it is handed."""
    P, tuning = pkg("PGCN"), pkg("tuning")
    monkeypatch.setattr(P, "_dense_stream", lambda t: None)
    monkeypatch.setattr(P, "_dense_lib", lambda: emu)
    monkeypatch.setattr(P, "_layer_fused_ok", lambda A, H, w: True)
    monkeypatch.setattr(P, "_layer_fused_level", lambda: 2)
    torch.manual_seed(4)
    n, f = 70, 32
    A = (torch.rand(n, n) < 0.1).float() * torch.rand(n, n)
    eng = _FakeEngine(A)
    l1, l2 = P.PGCN(eng, f, f), P.PGCN(eng, f, f)
    H0 = torch.randn(n, f)
    coef = torch.randn(n, f)
    premasks = []
    real_tb = torch.ops.aten.threshold_backward
    H = H0.clone().requires_grad_(True)
    out = l2(l1(H))
    import unittest.mock as mock
    with mock.patch.object(P.torch.ops.aten, "threshold_backward", side_effect=lambda *a: (premasks.append(1), real_tb(*a))[1]):
        (out * coef).sum().backward()
    Hr = H0.clone().requires_grad_(True)
    w1, w2 = l1.linear.weight.detach().clone().requires_grad_(True), l2.linear.weight.detach().clone().requires_grad_(True)
    ref = torch.relu((A @ torch.relu((A @ Hr) @ w1.t())) @ w2.t())
    (ref * coef).sum().backward()
    for got, want in ((out, ref), (H.grad, Hr.grad), (l1.linear.weight.grad, w1.grad), (l2.linear.weight.grad, w2.grad)):
        assert float((got - want).detach().abs().max()) <= 5e-6 * float(want.detach().abs().max())
    assert eng.calls == ["f", "f", "b", "b"]
    assert len(premasks) == 1, "only the top layer runs a separate mask pass; the lower one is handed a pre-masked gradient"


def test_cpu_tensors_never_reach_the_kernels():
    P, tuning = pkg("PGCN"), pkg("tuning")
    assert tuning.Tuning().dense_fused == 2 and tuning.Tuning().layer_fused == 0   # (tuning.py has the epochs behind both)
    assert P.linear_relu_fused(torch.randn(8, 8), torch.randn(4, 8)) is None
    assert P.linear_relu_grad_input_fused(torch.randn(8, 4), torch.randn(8, 4), torch.randn(4, 8)) is None


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
    gm, gx = P.linear_relu_grad_input_fused(g, y, w)
    assert torch.equal(gm, torch.ops.aten.threshold_backward(g, y, 0.0))
    exact = gm.double() @ w.double()
    e_mine, e_stock = float((gx.double() - exact).abs().max()), float(((gm @ w).double() - exact).abs().max())
    assert e_mine <= max(2 * e_stock, 1e-6 * float(exact.abs().max())), (e_mine, e_stock)
    assert torch.equal(P.linear_relu_grad_input_fused(g, y, w)[1], gx)
    xp = torch.randn(5000, 2 * f, device=dev)[:, :f]                   # leading dimension 256
    assert torch.allclose(P.linear_relu_fused(xp, w), (xp @ w.t()).clamp_min_(0), atol=1e-4)
    assert P.linear_relu_fused(torch.randn(64, 132, device=dev), torch.randn(8, 132, device=dev)) is None


@pytest.mark.gpu
def test_layer_with_the_kernels_switched_on(monkeypatch):
    """The autograd node of PGCN.py:146-147 with tuning.dense_fused = 2 against level 0 on the GPU."""
    P, tuning = pkg("PGCN"), pkg("tuning")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(2)
    x0, w0 = torch.randn(30011, 128, device=dev), torch.randn(128, 128, device=dev) / 11
    coef = torch.randn(30011, 128, device=dev)
    out = {}
    for level in (0, 2, 2):
        monkeypatch.setattr(tuning.T, "dense_fused", level)
        x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
        y = P._LinearReluNoBias.apply(x, w)
        (y * coef).sum().backward()
        torch.cuda.synchronize()
        if level in out:            # reproducible (the two tensors these kernels write; dW stays the library's batched product)
            assert torch.equal(out[level][0], y.detach()) and torch.equal(out[level][1], x.grad)
        out[level] = (y.detach(), x.grad, w.grad)
    for a, b in zip(out[2], out[0]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale


@pytest.mark.gpu
@pytest.mark.parametrize("n,fin,fout", FIX_SHAPES + [(232965, 128, 128), (100003, 64, 64)])
def test_fixup_loader_kernel(n, fin, fout):
    """pgcn_fixup_linear_f32 on the GPU: bit-identical to the ordered sums followed by the product kernel, S written out exactly,
    mask epilogue exact; both slot-list forms."""
    P = pkg("PGCN")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    if n > 5000:                                       # (the host-side reference sum of _random_deferred is a Python loop)
        g0 = torch.Generator().manual_seed(3)
        cnt = torch.randint(0, 7, (n,), generator=g0)
        cnt[::97] = 19
        direct = torch.rand(n, generator=g0) < 0.1
        cnt[direct] = 0
        total = int(cnt.sum())
        ws = torch.randn(total + 3, fin, generator=g0).to(dev)
        slot_ids = torch.randperm(total + 3, generator=g0)[:total].to(torch.int32).to(dev)
        begin = (torch.cumsum(cnt, 0) - cnt)
        base = torch.randn(n, fin, generator=g0).to(dev)
        row_fix = torch.stack([begin, torch.where(direct, torch.full_like(cnt, -1), cnt)], 1).to(torch.int32).contiguous().to(dev)
        # the ordered sums by the library's own fix-up kernel (csrc): the route the folded loader replaces
        K = pkg("kernels").HipKernels(dev)
        rows = torch.nonzero(~direct).reshape(-1)
        fix = torch.stack([rows, begin[rows], cnt[rows], torch.zeros_like(rows)], 1).to(torch.int32).contiguous().to(dev)
        S = base.clone()
        lib = pkg("_lib")
        lib.check(K.lib.pgcn_spmm_fixup_f32(fix.data_ptr(), fix.shape[0], slot_ids.data_ptr(), None, ws.data_ptr(), S.data_ptr(), fin, fin,
                                            0, st), "pgcn_spmm_fixup_f32")
        w = (torch.randn(fout, fin, generator=g0) / 8).to(dev)
        y, _ = P.fixup_linear_call(P._dense_lib(), row_fix, slot_ids, ws, base, fin, w, True, P.EPI_RELU, None, False, st)
        assert torch.equal(y, P.linear_relu_call(P._dense_lib(), S, w, True, st))
        w2 = (torch.randn(fin, fout, generator=g0) / 8).to(dev)
        m = torch.randn(n, fout, generator=g0).to(dev)
        y2, S2 = P.fixup_linear_call(P._dense_lib(), row_fix, slot_ids, ws, base, fin, w2, False, P.EPI_MASK, m, True, st)
        assert torch.equal(S2, S)
        assert torch.equal(y2, P.linear_epilogue_call(P._dense_lib(), S, w2, False, P.EPI_MASK, m, st))
        assert torch.equal(y2, torch.where(m > 0, P.linear_epilogue_call(P._dense_lib(), S, w2, False, P.EPI_NONE, None, st),
                                           torch.zeros((), device=dev)))
        torch.cuda.synchronize()
        return
    assert _check_fixup(P, P._dense_lib(), n, fin, fout, st, dev=dev) <= BOUND
    assert _check_fixup(P, P._dense_lib(), n, fin, fout, st, dev=dev, ids=False) <= BOUND
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("f", [128, 64])
def test_fused_layer_on_the_engine(f, monkeypatch):
    """tuning.layer_fused on the real engine (a graph with strips, bf16 blocks and a gather part): the forward is BIT-IDENTICAL to
    PSpMM + the separate product (same partial rows, same order of the sums, same product kernel), the re-associated backward agrees
    with it to fp32 rounding, twice the same bits, and the deferred sum can still be finished by the separate fix-up."""
    P, tuning, partition, synth = pkg("PGCN"), pkg("tuning"), pkg("partition"), pkg("synth")
    engine, kernels = pkg("engine"), pkg("kernels")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    n = 40000
    _, row, col, val = synth.make_graph(n, 6_000_000, seed=3)
    monkeypatch.setattr(partition, "CORE_MIN_NNZ", 0)
    monkeypatch.setattr(partition, "DENSE3_MIN_BLOCKS", 0)
    monkeypatch.setattr(partition, "STRIP_MIN_RECORDS", 0)
    part = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64), 0, 1)
    K = kernels.HipKernels(dev)
    eng = engine.AggregationEngine(part, K, dev)
    assert eng.A_loc.fix_all is not None, "the test graph should have tiled parts (a separate fix-up to fold)"
    P.device, P.myrank, P.world_size = dev, 0, 1
    P.init_stats()
    torch.manual_seed(0)
    H0 = torch.rand(n, f, device=dev)
    d = eng.forward_deferred(H0)
    assert isinstance(d, kernels.DeferredSum)
    assert torch.equal(d.finish(), eng.forward(H0))                      # the deferred sum completed by the separate fix-up
    l1, l2 = P.PGCN(eng, f, f).to(dev), P.PGCN(eng, f, f).to(dev)
    coef = torch.randn(n, f, device=dev)
    res = {}
    for level in (0, 2, 2, 1):
        monkeypatch.setattr(tuning.T, "layer_fused", level)
        for l in (l1, l2):
            l.linear.weight.grad = None
        H = H0.clone().requires_grad_(True)
        out = l2(l1(H))
        (out * coef).sum().backward()
        torch.cuda.synchronize()
        got = (out.detach(), H.grad, l1.linear.weight.grad.clone(), l2.linear.weight.grad.clone())
        if level in res:
            for a, b in zip(got, res[level]):
                assert torch.equal(a, b), "the fused layer is not reproducible"
        res[level] = got
    assert torch.equal(res[2][0], res[0][0]), "folding the fix-up into the product changed the forward's bits"
    for a, b in zip(res[1], res[2]):
        assert torch.equal(a, b), "the node with finished operands (level 1) and with the folded fix-up (level 2) differ"
    for a, b in zip(res[2][1:], res[0][1:]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), float((a - b).abs().max()) / float(b.abs().max())
