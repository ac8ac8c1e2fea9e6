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


class PlanKernels(OracleKernels):
    """The same interface with every SpMM executed FROM ITS LAUNCH PLAN (tests/plan_interpreter.py): ``prepare`` runs the
    host half of ``HipKernels`` (tasks, strip / core / MFMA tiles, slot lists) and ``spmm`` walks those arrays the way
    the kernels are specified to.  Slower than OracleKernels; used where a whole run should exercise the plans."""
    name = "plans(test-only)"

    def __init__(self):
        from plan_interpreter import HostPlanner
        self._planner = HostPlanner()

    def prepare(self, csr, pattern_only=False):
        return self._planner.prepare(csr, pattern_only)

    def spmm(self, A, B, C, accumulate=False):
        from plan_interpreter import run_plan
        if A.nrows == 0:
            return C
        if A.col.numel() == 0 and A.core is None and A.strip is None:
            if not accumulate:                       # an empty block: C = 0 (kernels._bind_spmm does the same without a launch)
                (C[:A.nrows] if A.row_map is None else C[A.row_map.long()]).zero_()
            return C
        out, _ = run_plan(A, B.detach().numpy(), C0=C.detach().numpy() if accumulate else None, accumulate=accumulate)
        rows = np.arange(A.nrows) if A.row_map is None else A.row_map.numpy().astype(np.int64)
        assert not np.isnan(out[rows]).any(), "the plan leaves output rows unwritten"
        C[torch.from_numpy(rows)] = torch.from_numpy(out[rows].astype(np.float32))
        return C


# ---- GAT path: numpy stand-ins for pgcn_gat.hip on the STORAGE order of the structure ----------
class _CpuGat(_CpuCSR):
    def __init__(self, csr, rows_wave, rows_block):
        super().__init__(csr)
        self.rowptr = csr.rowptr.cpu().numpy().astype(np.int64)
        self.col = csr.col.cpu().numpy().astype(np.int64)
        self.seg = np.repeat(np.arange(self.nrows), np.diff(self.rowptr))
        listed = np.sort(np.concatenate([rows_wave.cpu().numpy(), rows_block.cpu().numpy()]))
        assert np.array_equal(listed, np.arange(self.nrows)), "every row must be in exactly one list"


def _prepare_gat(self, csr, rows_wave, rows_block, chunk=None, small_row=None):
    assert csr.core is None and csr.row_map is None
    return _CpuGat(csr, rows_wave, rows_block)


def _with_values(self, A, plane):
    import copy
    B = copy.copy(A)
    B.v = plane.numpy()[:A.col.shape[0]]              # shares memory with the tensor: later writes are seen
    return B


def _raw(A, s1, s2, k):
    return s1.numpy()[A.seg, k] + s2.numpy()[A.col, k]


def _gat_edge_softmax(self, A, s1, s2, heads, slope, mode, n_global, alpha, beta, rowstat=None):
    nz = A.col.shape[0]
    ne = np.diff(A.rowptr) > 0
    starts = A.rowptr[:-1][ne]
    for k in range(heads):
        raw = _raw(A, s1, s2, k).astype(np.float32)
        m = np.zeros(A.nrows, np.float32)
        if mode == 0:
            raw = np.where(raw > 0, raw, raw * np.float32(slope))
            if nz:
                m[ne] = np.maximum.reduceat(raw, starts)
            em = np.zeros(A.nrows, np.float32)
        else:
            if nz:
                m[ne] = np.maximum(np.maximum.reduceat(raw, starts), 0)
            em = np.exp(-m)
        w = np.exp(raw - m[A.seg])
        D = (np.float32(n_global) - np.diff(A.rowptr).astype(np.float32)) * em
        if nz:
            D[ne] += np.add.reduceat(w, starts)
        inv = np.where(D > 0, 1 / np.where(D > 0, D, 1), 0).astype(np.float32)
        if alpha is not None:                              # (None: statistics only, like the device kernel)
            alpha[k, :nz] = torch.from_numpy((w - em[A.seg]) * inv[A.seg])
        if mode == 1:
            beta[:, k] = torch.from_numpy(em * inv)
        if rowstat is not None:
            rowstat[:, k, :] = torch.from_numpy(np.stack([s1.numpy()[:, k], m, inv, em], 1).astype(np.float32))


def _gat_edge_weights_t(self, AT, s2, rowstat, heads, slope, mode, alpha_t):
    nz = AT.col.shape[0]
    rs = rowstat.numpy()
    for k in range(heads):
        st = rs[AT.col, k]                               # (s1, m, 1/D, exp(-m)) of the ORIGINAL row = column here
        raw = st[:, 0] + s2.numpy()[AT.seg, k]
        if mode == 0:
            raw = np.where(raw > 0, raw, raw * np.float32(slope))
        alpha_t[k, :nz] = torch.from_numpy(((np.exp(raw - st[:, 1]) - st[:, 3]) * st[:, 2]).astype(np.float32))


def _gat_edge_grad(self, A, s1, s2, alpha, beta, Z, dOut, t, heads, d, slope, mode, de, ds1):
    nz = A.col.shape[0]
    Zn, Gn = Z.numpy(), dOut.numpy()
    for k in range(heads):
        dp = (Gn[A.seg, k * d:(k + 1) * d] * Zn[A.col, k * d:(k + 1) * d]).sum(1)
        p = alpha.numpy()[k, :nz] + (beta.numpy()[A.seg, k] if mode == 1 else 0)
        g = p * (dp - t.numpy()[A.seg, k])
        if mode == 0:
            g = g * np.where(_raw(A, s1, s2, k) > 0, 1.0, slope)
        de[:nz, k] = torch.from_numpy(g.astype(np.float32))          # de is entry-major [nnz, heads]
        acc = np.zeros(A.nrows, np.float32)
        np.add.at(acc, A.seg, g.astype(np.float32))
        ds1[:, k] = torch.from_numpy(acc)


def _csr_row_sums(self, A, perm, src, planes, out):
    nz = A.col.shape[0]
    for k in range(planes):
        v = src.numpy()[:nz, k]                                        # entry-major source
        if perm is not None:
            v = v[perm.numpy()]
        acc = np.zeros(A.nrows, np.float32)
        np.add.at(acc, A.seg, v)
        out[:A.nrows, k] = torch.from_numpy(acc)


def _csr_permute(self, src, perm, dst):
    n = perm.numel()
    if n:
        dst[:, :n] = src[:, perm]


def _gat_edge_grad_sliced(self, *a, **k):
    return False                                         # no sliced variant in the stand-in: the caller falls back


# ---- the one-launch multi-head products and the fused passes (pgcn_spmm_heads.hip), same shape rules ----------
def _heads_ok(heads, d):
    return heads <= 8 and heads * d <= 256 and d % 4 == 0


def _fused_ok(heads, d):
    hl = d // 4
    return _heads_ok(heads, d) and hl >= 8 and hl & (hl - 1) == 0


def _csr(A, w):
    import scipy.sparse as sp
    return sp.csr_matrix((w.astype(np.float64), A.col, A.rowptr), shape=(A.nrows, A.ncols))


def _put(C, cols, val, accumulate):
    t = torch.from_numpy(np.ascontiguousarray(val).astype(np.float32))
    if accumulate:
        C[:t.shape[0], cols] += t
    else:
        C[:t.shape[0], cols] = t


def _spmm_heads(self, A, alpha, B, C, heads, d, accumulate=False):
    if not _heads_ok(heads, d):
        return False
    nz, Bn = A.col.shape[0], B.detach().numpy().astype(np.float64)
    for k in range(heads):
        _put(C, slice(k * d, (k + 1) * d), _csr(A, alpha.numpy()[k, :nz]) @ Bn[:, k * d:(k + 1) * d], accumulate)
    return True


def _recomputed(st, s2v, slope, mode):
    """(alpha, raw) of entries whose softmax row has statistics ``st`` (s1, m, 1/D, exp(-m)) and whose other end has s2v."""
    raw = st[:, 0] + s2v
    r = np.where(raw > 0, raw, raw * np.float32(slope)) if mode == 0 else raw
    return ((np.exp(r - st[:, 1]) - st[:, 3]) * st[:, 2]).astype(np.float32), raw


def _spmm_heads_recompute(self, AT, rowstat, s2, slope, mode, B, C, heads, d, accumulate=False):
    if not _heads_ok(heads, d):
        return False
    Bn = B.detach().numpy().astype(np.float64)
    for k in range(heads):
        w, _ = _recomputed(rowstat.numpy()[AT.col, k], s2.numpy()[AT.seg, k], slope, mode)
        _put(C, slice(k * d, (k + 1) * d), _csr(AT, w) @ Bn[:, k * d:(k + 1) * d], accumulate)
    return True


def _spmm_heads_forward2(self, A, rowstat, s2, slope, mode, B, C, C2, heads, d, accumulate=False):
    """out = A_alpha . B with alpha from the statistics of the entry's ROW and s2 of its column; C2 = [V | C | 0]."""
    if not _fused_ok(heads, d):
        return False
    F, Bn = heads * d, B.detach().numpy().astype(np.float64)
    for k in range(heads):
        st = rowstat.numpy()[A.seg, k]
        w, raw = _recomputed(st, s2.numpy()[A.col, k], slope, mode)
        c = w * np.where(raw > 0, 1.0, slope).astype(np.float32) if mode == 0 else w + st[:, 3] * st[:, 2]
        _put(C, slice(k * d, (k + 1) * d), _csr(A, w) @ Bn[:, k * d:(k + 1) * d], accumulate)
        _put(C2, slice(k * d, (k + 1) * d), _csr(A, c) @ Bn[:, k * d:(k + 1) * d], accumulate)
        _put(C2, slice(F + k, F + k + 1), np.asarray(_csr(A, c).sum(1)), accumulate)
    pw2 = F + (heads + 3) // 4 * 4
    if not accumulate:
        C2[:A.nrows, F + heads:pw2] = 0
    return True


def _spmm_heads_grad(self, AT, rowstat, s2, slope, mode, B, Z, t, C, de, heads, d, accumulate=False):
    """On the TRANSPOSED structure (row j, col i): dZ_j = sum_i alpha_ij dOut_i, de_ij from <dOut_i, Z_j>, ds2_j beside it."""
    if not _fused_ok(heads, d):
        return False
    F = heads * d
    Bn, Zn = B.detach().numpy().astype(np.float64), Z.detach().numpy().astype(np.float64)
    for k in range(heads):
        st = rowstat.numpy()[AT.col, k]
        w, raw = _recomputed(st, s2.numpy()[AT.seg, k], slope, mode)
        c = w * np.where(raw > 0, 1.0, slope).astype(np.float32) if mode == 0 else w + st[:, 3] * st[:, 2]
        dp = (Bn[AT.col, k * d:(k + 1) * d] * Zn[AT.seg, k * d:(k + 1) * d]).sum(1)
        g = c * (dp - t.numpy()[AT.col, k])
        _put(C, slice(k * d, (k + 1) * d), _csr(AT, w) @ Bn[:, k * d:(k + 1) * d], accumulate)
        _put(C, slice(F + k, F + k + 1), np.asarray(_csr(AT, g).sum(1)), accumulate)
        if de is not None:
            de[:AT.col.shape[0], k] = torch.from_numpy(g.astype(np.float32))
    if not accumulate:
        C[:AT.nrows, F + heads:F + (heads + 3) // 4 * 4] = 0
    return True


OracleKernels.spmm_heads = _spmm_heads
OracleKernels.spmm_heads_recompute = _spmm_heads_recompute
OracleKernels.spmm_heads_forward2 = _spmm_heads_forward2
OracleKernels.spmm_heads_grad = _spmm_heads_grad
OracleKernels.gat_edge_grad_sliced = _gat_edge_grad_sliced
OracleKernels.prepare_gat = _prepare_gat
OracleKernels.with_values = _with_values
OracleKernels.gat_edge_softmax = _gat_edge_softmax
OracleKernels.gat_edge_grad = _gat_edge_grad
OracleKernels.gat_edge_weights_t = _gat_edge_weights_t
OracleKernels.csr_row_sums = _csr_row_sums
OracleKernels.csr_permute = _csr_permute
