"""The dense products of a layer by rocBLAS solution index (gemm/pgcn_gemm.cpp, PGCN.mm_nt / mm_nn): the library loads
and exports what include/pgcn_gemm.h declares, the TunableOp result file is read the way PyTorch writes it, records
for another rocBLAS build or GPU are ignored, and -- on the GPU -- the replayed kernels compute the same products as
PyTorch's default pick and a refused index falls back to it."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg


def test_gemm_library_exports_its_header():
    P = pkg("PGCN")
    src = open(os.path.join(ROOT, "include", "pgcn_gemm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(pgcn_gemm_[a-z0-9_]+)\s*\(", src)))
    assert names == ["pgcn_gemm_f32", "pgcn_gemm_last_error", "pgcn_gemm_rocblas_version", "pgcn_gemm_set_atomics"]
    L = ctypes.CDLL(P.GEMM_LIB_PATH)
    for n in names:
        assert hasattr(L, n)
    buf = ctypes.create_string_buffer(256)
    L.pgcn_gemm_rocblas_version.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    assert L.pgcn_gemm_rocblas_version(buf, 256) == 0 and re.match(rb"\d+\.\d+\.\d+", buf.value)
    assert L.pgcn_gemm_rocblas_version(buf, 2) != 0          # a buffer that cannot hold it is an error, not an overrun


def test_shipped_records_are_read_like_pytorch_writes_them(tmp_path):
    P = pkg("PGCN")
    val = dict(l.strip().split(",")[1:3] for l in open(P.TUNABLEOP_SHIPPED) if l.startswith("Validator,"))
    t = P.parse_tunableop_rocblas(P.TUNABLEOP_SHIPPED, val["ROCBLAS_VERSION"], val["GCN_ARCH_NAME"])
    # the benchmark shape: forward (tn) and input gradient (nn) of an n x 128 x 128 layer, keys as PyTorch builds them
    assert isinstance(t[("tn", 128, 232965, 128, 128, 128, 128)], int) and isinstance(t[("nn", 128, 232965, 128, 128, 128, 128)], int)
    assert all(k[0] in ("tn", "nn", "nt", "tt") and len(k) == 7 for k in t)
    # hipBLASLt choices and batched products are not replayed
    assert len(t) == sum(1 for l in open(P.TUNABLEOP_SHIPPED) if l.startswith("GemmTunableOp_float_") and "Gemm_Rocblas_" in l)
    # another rocBLAS build / another GPU: solution indices mean nothing there
    assert P.parse_tunableop_rocblas(P.TUNABLEOP_SHIPPED, "9.9.9", val["GCN_ARCH_NAME"]) == {}
    assert P.parse_tunableop_rocblas(P.TUNABLEOP_SHIPPED, val["ROCBLAS_VERSION"], "gfx942:sramecc+:xnack-") == {}
    assert P.parse_tunableop_rocblas(str(tmp_path / "absent.csv"), val["ROCBLAS_VERSION"], val["GCN_ARCH_NAME"]) == {}
    junk = tmp_path / "junk.csv"
    junk.write_text("Validator,ROCBLAS_VERSION,%s\nValidator,GCN_ARCH_NAME,%s\nGemmTunableOp_float_NN,nn_1_2,Gemm_Rocblas_7,0.1\n"
                    "GemmTunableOp_float_NN,nn_4_5_6_ld_4_6_4,Gemm_Rocblas_x,0.1\nGemmTunableOp_float_NN,nn_4_5_6_ld_4_6_4,Gemm_Rocblas_-11,0.1\n"
                    % (val["ROCBLAS_VERSION"], val["GCN_ARCH_NAME"]))
    assert P.parse_tunableop_rocblas(str(junk), val["ROCBLAS_VERSION"], val["GCN_ARCH_NAME"]) == {("nn", 4, 5, 6, 4, 6, 4): -11}


def test_products_on_cpu_tensors_are_pytorchs():
    P = pkg("PGCN")
    x, w = torch.randn(50, 8), torch.randn(6, 8)
    assert torch.equal(P.mm_nt(x, w), x @ w.t())
    g = torch.randn(50, 6)
    assert torch.equal(P.mm_nn(g, w), g @ w)


@pytest.mark.gpu
def test_replayed_kernels_compute_the_products_and_refusals_fall_back():
    P = pkg("PGCN")
    dev = torch.device("cuda:0")
    table = P._gemm_direct_table()
    assert table, "no rocBLAS records usable on this box (library missing, or tunableop/gfx950.csv is for another stack)"
    n, f = 232965, 128
    assert ("tn", f, n, f, f, f, f) in table and ("nn", f, n, f, f, f, f) in table
    g0 = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(n, f, generator=g0).to(dev)
    w = (torch.randn(f, f, generator=g0) / 11).to(dev)
    for name, fn, ref in (("tn", P.mm_nt, lambda: x @ w.t()), ("nn", P.mm_nn, lambda: x @ w)):
        direct = P._gemm_direct_call(name, w, x, f, n, f)
        assert direct is not None, "the %s record was refused" % name
        want = ref()
        exact = (x.double() @ (w.double().t() if name == "tn" else w.double()))
        e_direct, e_torch = float((direct.double() - exact).abs().max()), float((want.double() - exact).abs().max())
        assert e_direct <= max(2 * e_torch, 1e-5 * float(exact.abs().max())), (name, e_direct, e_torch)
        assert torch.equal(fn(x, w), direct)                 # the public entry takes the same route, reproducibly
    # a row-strided operand has another key: no record, PyTorch's product
    xs = torch.randn(n, 2 * f, device=dev)[:, :f]
    assert P._gemm_direct_call("tn", w, xs, f, n, f) is None and torch.allclose(P.mm_nt(xs, w), xs @ w.t(), atol=1e-4)
    # an index rocBLAS refuses: the record is dropped and the product is PyTorch's from then on
    key = ("tn", f, n, f, f, f, f)
    good = table[key]
    table[key] = 123456789
    try:
        y = P.mm_nt(x, w)
        assert key not in table and torch.allclose(y, x @ w.t(), atol=1e-4)
    finally:
        table[key] = good
    # through autograd: the layer of PGCN.py:146-147 with and without the records
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    y = P._LinearReluNoBias.apply(xg, wg)
    y.square().sum().backward()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    torch.relu(xr @ wr.t()).square().sum().backward()
    assert torch.allclose(xg.grad, xr.grad, rtol=1e-4, atol=1e-4) and torch.allclose(wg.grad, wr.grad, rtol=1e-3, atol=1e-2)
