"""Checker-backed kernel provider for the CPU-only tests (NOT part of the product).

Implements the ``kernels.HipKernels`` interface on CPU torch tensors by calling the
oracle (oracle/pgcn_oracle.c), so the host logic of the package -- partition layout,
exchange ordering over gloo, PSpMM forward/backward structure, statistics -- can be
exercised without a GPU.  It is injected by tests only (``PGCN._kernel_provider``)."""
import numpy as np
import torch

from oracle import oracle


class _CpuCSR:
    """Whole block (gather part + dense core) as one scipy CSR over OUTPUT rows."""

    def __init__(self, csr):
        import scipy.sparse as sp
        self.nrows, self.ncols, self.nnz = csr.nrows, csr.ncols, csr.nnz
        r, c, v = csr.to_coo()
        self.r, self.c, self.v = r.cpu().numpy(), c.cpu().numpy(), v.cpu().numpy().astype(np.float32)
        self.touched = None if csr.row_map is None else csr.row_map.cpu().numpy().astype(np.int64)

    def alg_bytes(self, f, a=None, b=None):
        return 0


class OracleKernels:
    name = "oracle(test-only)"

    def prepare(self, csr, pattern_only=False):
        return _CpuCSR(csr)

    def spmm(self, A, B, C, accumulate=False):
        import scipy.sparse as sp
        if A.nrows == 0:
            return C
        R = C.shape[0]
        M = sp.csr_matrix((A.v, (A.r, A.c)), shape=(R, A.ncols))
        M.sort_indices()
        out = torch.from_numpy(oracle.spmm(M, B.detach().numpy()))
        rows = torch.arange(A.nrows) if A.touched is None else torch.from_numpy(A.touched)
        if accumulate:
            C[rows] += out[rows]
        else:
            C[rows] = out[rows]
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
