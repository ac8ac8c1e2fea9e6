"""Checker-backed kernel provider for the CPU-only tests (NOT part of the product).

Implements the ``kernels.HipKernels`` interface on CPU torch tensors by calling the
oracle (oracle/pgcn_oracle.c), so the host logic of the package -- partition layout,
exchange ordering over gloo, PSpMM forward/backward structure, statistics -- can be
exercised without a GPU.  It is injected by tests only (``PGCN._kernel_provider``)."""
import numpy as np
import torch

from oracle import oracle


class _CpuCSR:
    def __init__(self, csr):
        self.nrows, self.ncols, self.nnz = csr.nrows, csr.ncols, csr.nnz
        self.rowptr = csr.rowptr.cpu().numpy().astype(np.int64)
        self.col = csr.col.cpu().numpy().astype(np.int32)
        self.val = csr.val.cpu().numpy().astype(np.float32)
        self.row_map = None if csr.row_map is None else csr.row_map.cpu().numpy()

    def alg_bytes(self, f, a=None, b=None):
        return 0


class OracleKernels:
    name = "oracle(test-only)"

    def prepare(self, csr, pattern_only=False):
        return _CpuCSR(csr)

    def spmm(self, A, B, C, accumulate=False):
        if A.nrows == 0:
            return C
        out = oracle.spmm_csr(A.rowptr, A.col, A.val, B.detach().numpy())
        t = torch.from_numpy(out)
        if A.row_map is None:
            if accumulate:
                C[:A.nrows] += t
            else:
                C[:A.nrows] = t
        else:
            idx = torch.from_numpy(A.row_map.astype(np.int64))
            if accumulate:
                C[idx] += t
            else:
                C[idx] = t
        return C

    def gather_rows(self, H, idx, out):
        n = idx.numel()
        if n:
            out[:n] = torch.from_numpy(oracle.gather_rows(H.detach().numpy(), idx.numpy()))
        return out

    def scatter_rows(self, H, idx, src, accumulate):
        n = idx.numel()
        if n:
            h = H.numpy()
            oracle.scatter_rows(h, idx.numpy(), src[:n].contiguous().numpy(), accumulate)
        return H
