"""GAT aggregation over the 1D partition (BASELINE config 5).

The reference's ``PGAT.forward`` (/root/reference/GPU/PGAT.py:138-151) is dense: an n x n score
matrix ``z1 + z2^T`` masked with ``A > 0``, a row softmax over all n columns and a dense
``attention @ Z``.  Here the same layer runs on the stored entries of rank p's row block:

    forward   [Z | s2] boundary rows -> all-to-all-v -> panel Zc = [local rows ; halo rows]
              row statistics of the edge softmax(s1_i + s2_j) (pgcn_gat_edge_softmax_f32, alpha = NULL)
              ONE gather pass (pgcn_spmm_heads_forward2_f32, r03): out = A_alpha . Zc with alpha recomputed per entry,
                  and V_i = sum_j c_ij Z_j, C_i = sum_j c_ij (c = alpha x LeakyReLU' | alpha + beta) for the backward
              [shapes it does not cover: alpha planes from the softmax kernel, then pgcn_spmm_heads_f32 / one CSR SpMM
               per head with val = alpha plane k]
    backward  ONE gather pass over the transposed structure    (pgcn_spmm_heads_grad_f32, r03):
                  dZc[:, head k] = A_alpha_k^T . dOut[:, head k]  (weights recomputed from the row statistics)
                  de_ij = edge gradient from <dOut_i, Z_j> -- Z_j is the task's own row, dOut_i is gathered anyway
                  ds2 = row sums of de (leave with the row)
              ds1_i = sum_j de_ij = <dOut_i, V_i> - t_i C_i from the forward pass's second accumulator: de is never
                  stored [without it: row sums of de over the forward structure, pgcn_csr_row_sums_f32]
              [shapes the fused kernel does not cover: pgcn_gat_edge_grad_*_f32 (row i gathers Z_j), then the
               transposed product, then ds2 = row sums of de over the transposed structure]
              halo rows of [dZ | ds2] travel back to their owners and are ADDED (reverse all-to-all-v)

Two semantics (``mode``): "standard" = LeakyReLU + softmax over the neighbours, K heads
(concatenated); "reference" = the literal arithmetic of PGAT.py, where the n - deg non-edges of a
row take part in the softmax with logit 0: out_i = sum_edges alpha_ij Z_j + beta_i sum_all Z_j, the
global column sum being one small all-reduce.  In both modes P ranks compute exactly what one
process computes on the whole graph (owned rows).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from .engine import BoundaryExchange
from .partition import HostCSR, Partition, csr_from_coo, pick_nslices
from .tuning import T as _T

LONG_ROW = _T.gat_long_row   # rows above this get a 256-thread workgroup
MODES = {"standard": 0, "reference": 1}


@dataclass
class GatGraph:
    """Rank p's row block over the combined column space [local rows ; halo slab] -- forward
    structure, transposed structure, and the permutation between their storage orders."""
    n_local: int
    n_halo: int
    fwd: HostCSR                 # n_local x (n_local + n_halo), pattern
    bwd: HostCSR                 # (n_local + n_halo) x n_local, pattern
    perm: torch.Tensor           # int64 [nnz]: entry p of bwd is entry perm[p] of fwd
    fwd_wave: torch.Tensor       # int32 row lists (longest first)
    fwd_block: torch.Tensor
    bwd_wave: torch.Tensor
    bwd_block: torch.Tensor
    # r06: the transposed structure once more as its halo rows and its local rows (N > 1 only): the backward computes the halo rows
    # first and sends them home while the local rows are computed (GatEngine.backward)
    bwd_halo: Optional[HostCSR] = None       # n_halo x n_local
    bwd_local: Optional[HostCSR] = None      # n_local x n_local
    bwd_halo_lists: Optional[tuple] = None   # (wave rows, block rows) of each
    bwd_local_lists: Optional[tuple] = None
    # ... and the forward structure as its local columns and its halo columns: the row statistics need s2 only (a 16-byte-per-row
    # exchange), so the product over the local columns runs while the 1 KB rows of the halo are still on the wire
    fwd_local: Optional[HostCSR] = None      # n_local x n_local
    fwd_halo: Optional[HostCSR] = None       # n_local x n_halo
    fwd_local_lists: Optional[tuple] = None
    fwd_halo_lists: Optional[tuple] = None

    @property
    def nnz(self) -> int:
        return int(self.fwd.col.numel())


def _row_lists(rowptr: torch.Tensor, long_row: int):
    ln = (rowptr[1:] - rowptr[:-1])
    order = torch.argsort(-ln, stable=True)
    big = ln[order] > long_row
    return order[~big].to(torch.int32).contiguous(), order[big].to(torch.int32).contiguous()


def build_gat_graph(part: Partition, positive_only: bool = False, long_row: Optional[int] = None,
                    split_backward: bool = False) -> GatGraph:
    """Coalesce the partition's pieces into one pattern over [local ; halo] columns.
    Duplicate coordinates count once (the reference masks a DENSE matrix, PGAT.py:146);
    ``positive_only`` keeps the coordinates whose summed value is > 0 (``A > 0``)."""
    n_p, n_h = part.n_local, part.n_halo
    ncols = n_p + n_h
    rs, cs, vs = [], [], []
    r, c, v = part.A_loc.to_coo()
    rs.append(r.to(torch.int64)); cs.append(c.to(torch.int64)); vs.append(v)
    for a in part.A_halo:
        r, c, v = a.to_coo()
        rs.append(r.to(torch.int64)); cs.append(c.to(torch.int64) + n_p); vs.append(v)
    r, c, v = torch.cat(rs), torch.cat(cs), torch.cat(vs).to(torch.float64)
    key, inv = torch.unique(r * ncols + c, return_inverse=True)
    vsum = torch.zeros(key.numel(), dtype=torch.float64, device=key.device).index_add_(0, inv, v)
    if positive_only:
        key = key[vsum > 0]
    r, c = key // ncols, key % ncols
    ones = torch.ones(r.numel(), dtype=torch.float32, device=r.device)
    # forward storage order = the SpMM kernels' (row, col % S, col); pre-sorted so that
    # csr_from_coo's stable sort is the identity and positions are known
    S = pick_nslices(ncols)
    of = torch.argsort((r * S + c % S) * max(ncols, 1) + c, stable=True)
    r, c = r[of], c[of]
    fwd = csr_from_coo(r, c, ones, n_p, ncols, nslices=S, core=False)
    assert torch.equal(fwd.col.to(torch.int64), c)
    St = pick_nslices(n_p)
    perm = torch.argsort((c * St + r % St) * max(n_p, 1) + r, stable=True)
    bwd = csr_from_coo(c[perm], r[perm], ones, ncols, n_p, nslices=St, core=False)
    assert torch.equal(bwd.col.to(torch.int64), r[perm])
    lr = LONG_ROW if long_row is None else long_row
    fw, fb = _row_lists(fwd.rowptr, lr)
    bw, bb = _row_lists(bwd.rowptr, lr)
    g = GatGraph(n_p, n_h, fwd, bwd, perm.contiguous(), fw, fb, bw, bb)
    if n_h > 0 and split_backward:
        cb, rb = c[perm], r[perm]
        hal = cb >= n_p
        g.bwd_halo = csr_from_coo(cb[hal] - n_p, rb[hal], ones[:int(hal.sum())], n_h, n_p, nslices=St, core=False)
        g.bwd_local = csr_from_coo(cb[~hal], rb[~hal], ones[:int((~hal).sum())], n_p, n_p, nslices=St, core=False)
        g.bwd_halo_lists = _row_lists(g.bwd_halo.rowptr, lr)
        g.bwd_local_lists = _row_lists(g.bwd_local.rowptr, lr)
        hal = c >= n_p
        g.fwd_local = csr_from_coo(r[~hal], c[~hal], ones[:int((~hal).sum())], n_p, n_p, nslices=pick_nslices(n_p), core=False)
        g.fwd_halo = csr_from_coo(r[hal], c[hal] - n_p, ones[:int(hal.sum())], n_p, n_h, nslices=pick_nslices(n_h), core=False)
        g.fwd_local_lists = _row_lists(g.fwd_local.rowptr, lr)
        g.fwd_halo_lists = _row_lists(g.fwd_halo.rowptr, lr)
    return g


def _split_blocks(csr: HostCSR, kernels, chunk: int, small: int, long_row: int, row_bands=None, col_bands=None):
    """(gather structure of the remaining entries, device blocks) of an attention pattern, or None when too little of it is dense
    (``tuning.gat_block_tau``: a block pays from ~4 % fill on -- four heads of a block cost ~45 us of one CU, a gathered entry
    ~18 ns)."""
    from .partition import split_dense3
    if csr.nnz == 0 or csr.ncols < 128 or csr.nrows < 1:
        return None
    rows = torch.repeat_interleave(torch.arange(csr.nrows, dtype=torch.int64, device=csr.rowptr.device), csr.rowptr[1:] - csr.rowptr[:-1])
    cols = csr.col.to(torch.int64)
    ones = torch.ones(rows.numel(), dtype=torch.float32, device=rows.device)
    keep, h3 = split_dense3(rows, cols, ones, csr.nrows, csr.ncols, float(_T.gat_block_tau), row_bands, col_bands,
                            piece=int(_T.gat_block_piece) or None)
    if h3 is None or h3.nnz < _T.gat_block_min_frac * csr.nnz:
        return None
    rest = csr_from_coo(rows[keep], cols[keep], ones[:int(keep.sum())], csr.nrows, csr.ncols, nslices=csr.nslices, core=False)
    rest_dev = kernels.prepare_gat(rest, *_row_lists(rest.rowptr, long_row), chunk=chunk, small_row=small)
    return rest_dev, kernels.prepare_gat_blocks(h3)


@dataclass
class GatLayerState:
    """Per-layer buffers that live from forward to backward."""
    heads: int
    d: int
    beta: torch.Tensor                   # [n_local, heads]  (reference mode)
    rowstat: torch.Tensor                # [n_local, heads, 4] (s1, m, 1/D, exp(-m)) of the softmax rows
    alpha: Optional[torch.Tensor] = None     # [heads, nnz] head-major edge weights (the SpMM `val` planes): only for
    fwd_heads: Optional[List[object]] = None  # ... the shapes the recomputing kernels do not cover (GatEngine.planes)
    V: Optional[torch.Tensor] = None     # [n_local, F + heads (+pad)]: V_i | C_i of the two-accumulator forward product
    fused: bool = False                  # the last forward ran pgcn_spmm_heads_forward2_f32 (no alpha planes, no de)
    Zc: Optional[torch.Tensor] = None    # [(n_local + n_halo), Fp] = [Z | s2 | pad] of local and halo rows
    Zc_borrowed: bool = False            # Zc aliases the caller's packed projection (N = 1): a later copy-in forward must reallocate
    s2c: Optional[torch.Tensor] = None   # [(n_local + n_halo), heads] s2 of local and halo rows, compact (L2 resident)
    s1: Optional[torch.Tensor] = None
    out: Optional[torch.Tensor] = None
    busy: bool = False                   # a forward whose backward has not run yet owns these buffers


class GatEngine(BoundaryExchange):
    """Device-resident GAT structures of one rank + forward / backward of the aggregation."""

    def __init__(self, part: Partition, kernels, device: torch.device, exchanger=None, mode: str = "standard",
                 slope: float = 0.2, long_row: Optional[int] = None, overlap: Optional[bool] = None):
        # r06: the exchanges run on the comm stream like the GCN engine's (PGCN_OVERLAP=0 switches it off).  The forward needs the halo
        # rows of [Z | s2] before its one gather pass can start; the BACKWARD computes the halo rows of [dZ | ds2] first and sends them
        # home while the local rows are computed (PGAT.py:138-151; the overlap structure of Parallel-GCN/main.c:238-299)
        super().__init__(part, kernels, device, exchanger, overlap=overlap)
        if mode not in MODES:
            raise ValueError("mode must be 'standard' or 'reference', got %r" % (mode,))
        self.mode, self.mode_id, self.slope = mode, MODES[mode], float(slope)
        self.n_global = part.n
        g = build_gat_graph(part, positive_only=(mode == "reference"), long_row=long_row, split_backward=self.overlap)
        self.graph = g
        self.nnz = g.nnz
        # the provider's plan parameters are the GCN path's: the attention structures get their own (tuning.gat_chunk / gat_small_row),
        # passed as arguments -- nothing of the shared provider is touched
        chunk = max(int(getattr(kernels, "chunk", 0)), int(_T.gat_chunk))
        small = max(int(getattr(kernels, "small_row", 0)), int(_T.gat_small_row))
        self.fwd = kernels.prepare_gat(g.fwd, g.fwd_wave, g.fwd_block, chunk=chunk, small_row=small)
        self.bwd = kernels.prepare_gat(g.bwd, g.bwd_wave, g.bwd_block, chunk=chunk, small_row=small)
        self.bwd_halo = self.bwd_local = None
        if g.bwd_halo is not None:
            self.bwd_halo = kernels.prepare_gat(g.bwd_halo, *g.bwd_halo_lists, chunk=chunk, small_row=small)
            self.bwd_local = kernels.prepare_gat(g.bwd_local, *g.bwd_local_lists, chunk=chunk, small_row=small)
        self.fwd_local = self.fwd_halo = None
        if g.fwd_halo is not None:
            self.fwd_local = kernels.prepare_gat(g.fwd_local, *g.fwd_local_lists, chunk=chunk, small_row=small)
            self.fwd_halo = kernels.prepare_gat(g.fwd_halo, *g.fwd_halo_lists, chunk=chunk, small_row=small)
        # r06: the dense 512 x 128 blocks of the pattern on the bf16 matrix cores (pgcn_gat_blocks.hip) -- weights computed in registers from
        # the row / column statistics; the gather kernels keep the remaining entries.  Standard mode; every structure the fused passes walk
        # (the whole pattern, or its local / halo parts under the overlapped exchange) is split on its own: self.parts[name] =
        # (gather structure of the remaining entries, blocks) where enough of it is dense
        self.parts = {}
        self.blocks_nnz = 0
        if _T.gat_blocks and mode == "standard" and hasattr(kernels, "prepare_gat_blocks"):
            lr = LONG_ROW if long_row is None else long_row
            names = ("fwd_local", "fwd_halo", "bwd_local", "bwd_halo") if g.fwd_halo is not None else ("fwd", "bwd")
            # the block grid restarts at the bands of the vertex order (its communities) along every LOCAL index space, as the GCN blocks' does
            lb = getattr(part, "local_bands", None)
            lb_all = None if lb is None else torch.cat([lb.to(torch.int64).cpu(), torch.tensor([g.n_local], dtype=torch.int64)]) if g.n_halo else lb
            bands = {"fwd": (lb, lb_all), "bwd": (lb_all, lb), "fwd_local": (lb, lb), "bwd_local": (lb, lb), "fwd_halo": (lb, None),
                     "bwd_halo": (None, lb)}
            for name in names:
                sp = _split_blocks(getattr(g, name), kernels, chunk, small, lr, *bands[name])
                if sp is not None:
                    self.parts[name] = sp
            self.blocks_nnz = sum(self.parts[k][1].nnz for k in self.parts if k.startswith("fwd"))
        self.fwd_blocks = self.parts["fwd"][1] if "fwd" in self.parts else None      # (what bench.py and the tests report)
        self.perm = g.perm.to(self.device)
        self._inv_perm = None              # forward entry -> its position in the transposed structure (built on demand)
        self._scratch = {}
        self.sliced_grad = _T.gat_sliced   # XCD-sliced edge gradient where the shape allows
        # the edge gradient over the balanced tasks of the SpMM plan (pgcn_gat_edge_grad_tasks_f32)
        self.task_grad = _T.gat_task_grad and hasattr(kernels, "gat_edge_grad_tasks")
        # all heads of `attention @ Z` (and of its transpose) in one launch (pgcn_spmm_heads_f32)
        self.multi_head = _T.gat_multihead and hasattr(kernels, "spmm_heads")
        # the edge gradient inside the transposed product's gather pass (pgcn_spmm_heads_grad_f32)
        self.fused_grad = _T.gat_fused_grad and self.multi_head and hasattr(kernels, "spmm_heads_grad")
        # ... and the forward product with recomputed weights + the second accumulator that makes de unnecessary
        self.fused_fwd = _T.gat_fused_forward and self.fused_grad and hasattr(kernels, "spmm_heads_forward2")

    @property
    def inv_perm(self) -> torch.Tensor:
        if self._inv_perm is None:
            inv = torch.empty_like(self.perm)
            inv[self.perm] = torch.arange(self.perm.numel(), dtype=torch.int64, device=self.perm.device)
            self._inv_perm = inv
        return self._inv_perm

    # -- buffers ---------------------------------------------------------
    def _plane_scratch(self, name: str, heads: int) -> torch.Tensor:
        t = self._scratch.get((name, heads))
        if t is None:
            t = torch.empty((heads, max(self.nnz, 1)), dtype=torch.float32, device=self.device)
            self._scratch[(name, heads)] = t
        return t

    def new_layer_state(self, heads: int, d: int) -> GatLayerState:
        beta = torch.zeros((self.n_local, heads), dtype=torch.float32, device=self.device)
        rowstat = torch.zeros((self.n_local, heads, 4), dtype=torch.float32, device=self.device)
        return GatLayerState(heads, d, beta, rowstat)

    def planes(self, st: GatLayerState) -> torch.Tensor:
        """The alpha planes of ``st`` (4 bytes per entry and head), allocated when a path first needs them."""
        if st.alpha is None:
            st.alpha = torch.zeros((st.heads, max(self.nnz, 1)), dtype=torch.float32, device=self.device)
            st.fwd_heads = [self.k.with_values(self.fwd, st.alpha[k]) for k in range(st.heads)]
        return st.alpha

    def covers(self, heads: int, d: int) -> bool:
        """Shapes of the recomputing kernels (pgcn_spmm_heads_forward2_f32 / _grad_f32): 8, 16, 32 or 64 lanes per head."""
        hl = d // 4
        return d % 4 == 0 and hl >= 8 and hl & (hl - 1) == 0 and heads * d <= 256 and heads <= 8

    @staticmethod
    def padded_width(F: int, heads: int) -> int:
        return (F + heads + 3) // 4 * 4

    def _allreduce(self, buf: torch.Tensor) -> torch.Tensor:
        if self.size > 1:
            self.allreduce_sum(buf)          # BoundaryExchange: same communicator AND stream as the slabs
        return buf

    # -- one pass = the gather kernel over the entries outside the blocks + the blocks ------------------------------
    def _forward2(self, name: str, st: GatLayerState, s2, B, out, K: int, d: int, accumulate: bool = False) -> bool:
        A = getattr(self, name)
        part = self.parts.get(name) if d == 64 else None
        if not self.k.spmm_heads_forward2(part[0] if part else A, st.rowstat, s2, self.slope, self.mode_id, B, out, st.V, K, d,
                                          accumulate=accumulate):
            return False
        if part and not self.k.gat_blocks_forward(part[1], st.rowstat, s2, self.slope, B, out, st.V, K, d):
            raise RuntimeError("the block part of the GAT forward was refused after the gather part was taken")
        return True

    def _grad(self, name: str, st: GatLayerState, s2, dOut, Z, t, dZc, K: int, d: int) -> bool:
        AT = getattr(self, name)
        part = self.parts.get(name) if d == 64 else None
        if not self.k.spmm_heads_grad(part[0] if part else AT, st.rowstat, s2, self.slope, self.mode_id, dOut, Z, t, dZc, None, K, d):
            return False
        if part and not self.k.gat_blocks_backward(part[1], st.rowstat, s2, self.slope, dOut, Z, t, dZc, K, d):
            raise RuntimeError("the block part of the GAT backward was refused after the gather part was taken")
        return True

    # -- forward ---------------------------------------------------------
    def forward(self, st: GatLayerState, Z: torch.Tensor, s1: torch.Tensor, s2: torch.Tensor,
                panel: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``panel`` (optional): a matrix whose first F + K columns ARE [Z | s2] (PGAT's packed projection [Z | s2 | s1]); without
        halo rows it is used as the gather panel itself instead of being copied into one (r05: 239 MB per layer)."""
        K, d = st.heads, st.d
        F = K * d
        n_p, n_h = self.n_local, self.n_halo
        if Z.shape != (n_p, F) or s1.shape != (n_p, K) or s2.shape != (n_p, K):
            raise ValueError("expected Z %s, s1/s2 %s" % ((n_p, F), (n_p, K)))
        if panel is not None and n_h == 0 and panel.shape[0] == n_p and panel.shape[1] >= F + K and panel.shape[1] % 4 == 0 \
                and panel.stride(1) == 1 and panel.stride(0) == panel.shape[1] and panel.dtype is torch.float32 \
                and panel.data_ptr() % 16 == 0:
            st.Zc = Zc = panel.detach()
            st.Zc_borrowed = True              # an alias of the caller's (autograd-owned) projection: never written through
            Fp = Zc.shape[1]
        else:
            Fp = self.padded_width(F, K)
            if st.Zc is None or st.Zc_borrowed or st.Zc.shape != (n_p + n_h, Fp):
                st.Zc = torch.zeros((n_p + n_h, Fp), dtype=torch.float32, device=self.device)
                st.Zc_borrowed = False
            Zc = st.Zc
            Zc[:n_p, :F].copy_(Z)
            Zc[:n_p, F:F + K].copy_(s2)
        st.s1 = s1.contiguous()
        if st.s2c is None or st.s2c.shape != (n_p + n_h, K):
            # (zeros: under an emulated exchange -- bench.py --emulate-rank -- the halo rows are never written)
            st.s2c = torch.zeros((n_p + n_h, K), dtype=torch.float32, device=self.device)
        out = torch.empty((n_p, F), dtype=torch.float32, device=self.device)
        st.fused = False
        split = (self.size > 1 and self.overlap and n_h > 0 and self.fwd_halo is not None and self.fused_fwd and self.covers(K, d)
                 and self.mode_id == 0)
        if split:
            # N > 1 (r06): the statistics of the edge softmax need s2 only -- its halo rows (16 bytes each) are exchanged FIRST, the 1 KB
            # rows of [Z | s2] follow on the comm stream under the statistics pass and the product over the LOCAL columns; the product
            # over the halo columns (accumulated) waits for them.  PGAT.py:138-151; the overlap structure of Parallel-GCN/main.c:238-299
            st.s2c[:n_p].copy_(s2)
            send2 = self._slab("gat_send_s2", self.n_send, K)
            self.k.gather_rows(st.s2c[:n_p], self.send_idx, send2)
            w2 = self._exchange_all(send2, self.round_send_off, st.s2c[n_p:], self.round_recv_off, K, tag="forward_s2")
            send = self._slab("gat_send", self.n_send, Fp)
            self.k.gather_rows(Zc[:n_p], self.send_idx, send)
            wz = self._exchange_all(send, self.round_send_off, Zc[n_p:], self.round_recv_off, Fp, tag="forward")
            for w in w2:
                w()
            self.k.gat_edge_softmax(self.fwd, st.s1, st.s2c, K, self.slope, self.mode_id, self.n_global, None, st.beta, st.rowstat)
            pw2 = F + (K + 3) // 4 * 4
            if st.V is None or st.V.shape != (n_p, pw2):
                st.V = torch.empty((n_p, pw2), dtype=torch.float32, device=self.device)
            ok = self._forward2("fwd_local", st, st.s2c[:n_p], Zc[:n_p], out, K, d)
            for w in wz:
                w()
            if ok:
                ok = self._forward2("fwd_halo", st, st.s2c[n_p:], Zc[n_p:], out, K, d, accumulate=True)
            if not ok:
                raise RuntimeError("the split GAT forward was refused by a kernel whose shape `covers` accepted")
            st.fused = True
            st.out = out
            return out
        if self.size > 1:                                   # PGAT.py:139 `Comm.apply(H)`: here the rows of [Z | s2]
            send = self._slab("gat_send", self.n_send, Fp)
            self.k.gather_rows(Zc[:n_p], self.send_idx, send)
            for w in self._exchange_all(send, self.round_send_off, Zc[n_p:], self.round_recv_off, Fp):
                w()
        st.s2c.copy_(Zc[:, F:F + K])        # compact: the per-entry s2 gathers stay in L2 instead of striding the panel
        if self.fused_fwd and self.covers(K, d):
            # statistics only, then ONE gather pass: out, and V | C for the backward's ds1 (no alpha planes, no de)
            self.k.gat_edge_softmax(self.fwd, st.s1, st.s2c, K, self.slope, self.mode_id, self.n_global,
                                    None, st.beta, st.rowstat)
            pw2 = F + (K + 3) // 4 * 4
            if st.V is None or st.V.shape != (n_p, pw2):
                st.V = torch.empty((n_p, pw2), dtype=torch.float32, device=self.device)
            st.fused = self._forward2("fwd", st, st.s2c, Zc, out, K, d)
        if not st.fused:
            alpha = self.planes(st)
            self.k.gat_edge_softmax(self.fwd, st.s1, st.s2c, K, self.slope, self.mode_id, self.n_global,
                                    alpha, st.beta, st.rowstat)
            if not (self.multi_head and self.k.spmm_heads(self.fwd, alpha, Zc, out, K, d)):
                for k in range(K):            # shapes the one-launch kernel does not cover: one SpMM per head
                    self.k.spmm(st.fwd_heads[k], Zc[:, k * d:(k + 1) * d], out[:, k * d:(k + 1) * d])
        if self.mode_id == 1:                               # + beta_i * (sum over ALL vertices of Z_j)
            zsum = self._allreduce(Z.sum(0))
            out.view(n_p, K, d).addcmul_(st.beta.view(n_p, K, 1), zsum.view(1, K, d))
        st.out = out
        return out

    # -- backward --------------------------------------------------------
    def backward(self, st: GatLayerState, dOut: torch.Tensor, pack: Optional[torch.Tensor] = None):
        """Returns (dZ, ds1, ds2) for the owned rows; with ``pack`` (n_local x (F + 2K)) they are written there as
        [dZ | ds2 | ds1] instead (the gradient of PGAT's packed projection [Z | s2 | s1]) and ``pack`` is returned."""
        K, d = st.heads, st.d
        F = K * d
        n_p, n_h = self.n_local, self.n_halo
        Fp = st.Zc.shape[1]
        dOut = dOut.contiguous()
        dots = None
        if hasattr(self.k, "gat_row_dots") and dOut.stride(1) == 1 and st.out.stride(1) == 1:
            # t and (fused forward) ds1 in ONE pass over dOut, out and V (r05) instead of four element-wise / reduction launches
            dots = self.k.gat_row_dots(dOut, st.out, st.V if st.fused else None, K, d)
        t = dots[0] if dots is not None else (dOut.view(n_p, K, d) * st.out.view(n_p, K, d)).sum(-1).contiguous()
        if pack is not None and n_h == 0 and pack.shape == (n_p, Fp) and Fp >= F + 2 * K and pack.stride(0) == Fp:
            dZc = pack                 # the kernels write [dZ | ds2] straight into the gradient that goes back (no slab, no copy)
        else:
            dZc = self._slab("gat_dzc", n_p + n_h, Fp)[:n_p + n_h]
        if st.fused and self.bwd_halo is not None and self.overlap and n_h > 0:
            # N > 1 (r06): the same pass in two parts -- the HALO rows of [dZ | ds2] first, their way home on the comm stream under the
            # local rows' part
            if self._grad("bwd_halo", st, st.s2c[n_p:], dOut, st.Zc[n_p:], t, dZc[n_p:], K, d):
                if Fp > F + K:
                    dZc[n_p:, F + K:].zero_()
                back = self._slab("gat_send", self.n_send, Fp)
                waits = self._exchange_all(dZc[n_p:], self.round_recv_off, back, self.round_send_off, Fp, tag="backward")
                if not self._grad("bwd_local", st, st.s2c[:n_p], dOut, st.Zc[:n_p], t, dZc[:n_p], K, d):
                    raise RuntimeError("the local part of the fused GAT backward was refused after its halo part was taken")
                ds1 = dots[1] if dots is not None and dots[1] is not None else \
                    (dOut.view(n_p, K, d) * st.V[:, :F].view(n_p, K, d)).sum(-1) - t * st.V[:, F:F + K]
                return self._finish_backward(st, dOut, dZc, ds1, pack, waits=(back, waits))
        if st.fused:
            # one gather pass: dZc = A_alpha^T . dOut and ds2 = the row sums of the edge gradient, which is not stored:
            # ds1 = its column sums = <dOut_i, V_i> - t_i C_i from the forward pass's second accumulator
            if self._grad("bwd", st, st.s2c, dOut, st.Zc, t, dZc, K, d):
                ds1 = dots[1] if dots is not None and dots[1] is not None else \
                    (dOut.view(n_p, K, d) * st.V[:, :F].view(n_p, K, d)).sum(-1) - t * st.V[:, F:F + K]
                return self._finish_backward(st, dOut, dZc, ds1, pack)
            # the fused kernel refused operands its forward twin took (an alignment / stride of dOut or the work-space that
            # `covers` does not model): the unfused passes need the alpha planes the fused forward never wrote -- build them now
            self.k.gat_edge_softmax(self.fwd, st.s1, st.s2c, K, self.slope, self.mode_id, self.n_global,
                                    self.planes(st), st.beta, st.rowstat)
            st.fused = False
        if self.fused_grad:
            # one gather pass: dZc = A_alpha^T . dOut, de (entry-major, TRANSPOSED storage order) and ds2 = its row sums
            de_t = self._scratch.get(("de_t", K))
            if de_t is None:
                de_t = self._scratch[("de_t", K)] = torch.empty((max(self.nnz, 1), K), dtype=torch.float32, device=self.device)
            if self.k.spmm_heads_grad(self.bwd, st.rowstat, st.s2c, self.slope, self.mode_id, dOut, st.Zc, t, dZc, de_t, K, d):
                ds1 = torch.empty((n_p, K), dtype=torch.float32, device=self.device)
                self.k.csr_row_sums(self.fwd, self.inv_perm, de_t, K, ds1)
                return self._finish_backward(st, dOut, dZc, ds1, pack)
        de = self._scratch.get(("de", K))          # the edge gradient, ENTRY-major [nnz, K] (read once, through perm)
        if de is None:
            de = self._scratch[("de", K)] = torch.empty((max(self.nnz, 1), K), dtype=torch.float32, device=self.device)
        ds1 = None
        if self.task_grad:
            ds1t = torch.empty((n_p, K), dtype=torch.float32, device=self.device)
            if self.k.gat_edge_grad_tasks(self.fwd, st.s1, st.s2c, st.alpha, st.beta, st.Zc, dOut, t, K, d, self.slope,
                                          self.mode_id, de, ds1t):
                ds1 = ds1t
        ds1p = torch.empty((n_p, 8, K), dtype=torch.float32, device=self.device) if ds1 is None else None
        if ds1 is not None:
            pass
        elif self.sliced_grad and self.k.gat_edge_grad_sliced(self.fwd, st.s1, st.s2c, st.alpha, st.beta, st.Zc, dOut, t,
                                                            K, d, self.slope, self.mode_id, de, ds1p):
            ds1 = ds1p.sum(1)               # 8 per-XCD partials per row (plumbing: n_p x 8 x K floats)
        else:
            ds1 = torch.empty((n_p, K), dtype=torch.float32, device=self.device)
            self.k.gat_edge_grad(self.fwd, st.s1, st.s2c, st.alpha, st.beta, st.Zc, dOut, t, K, d,
                                 self.slope, self.mode_id, de, ds1)
        # dZc = A_alpha^T . dOut: the weights of the transposed structure are recomputed inside the gather kernel from
        # the row statistics (r03) -- or, for shapes / providers without that kernel, written as planes first
        if not (self.multi_head and hasattr(self.k, "spmm_heads_recompute")
                and self.k.spmm_heads_recompute(self.bwd, st.rowstat, st.s2c, self.slope, self.mode_id, dOut, dZc, K, d)):
            alpha_t = self._plane_scratch("alpha_t", K)
            self.k.gat_edge_weights_t(self.bwd, st.s2c, st.rowstat, K, self.slope, self.mode_id, alpha_t)
            bwd_heads = self._scratch.get(("bwd_heads", K))
            if bwd_heads is None:
                bwd_heads = [self.k.with_values(self.bwd, alpha_t[k]) for k in range(K)]
                self._scratch[("bwd_heads", K)] = bwd_heads
            if not (self.multi_head and self.k.spmm_heads(self.bwd, alpha_t, dOut, dZc, K, d)):
                for k in range(K):
                    self.k.spmm(bwd_heads[k], dOut[:, k * d:(k + 1) * d], dZc[:, k * d:(k + 1) * d])
        self.k.csr_row_sums(self.bwd, self.perm, de, K, dZc[:, F:F + K])
        return self._finish_backward(st, dOut, dZc, ds1, pack)

    def _finish_backward(self, st: GatLayerState, dOut: torch.Tensor, dZc: torch.Tensor, ds1: torch.Tensor,
                         pack: Optional[torch.Tensor] = None, waits=None):
        """Halo rows of [dZ | ds2] back to their owners (added), then the owned rows.  ``waits`` = (slab, waiters) of an exchange
        of the halo rows that is already on its way (the split backward)."""
        K, d = st.heads, st.d
        F = K * d
        n_p, n_h = self.n_local, self.n_halo
        Fp = st.Zc.shape[1]
        if Fp > F + K:
            (dZc[:n_p] if waits is not None else dZc)[:, F + K:].zero_()
        if self.size > 1:                                   # partial rows of [dZ | ds2] back to their owners, ADDED
            if waits is not None:
                back, waits = waits
            else:
                back = self._slab("gat_send", self.n_send, Fp)
                waits = self._exchange_all(dZc[n_p:], self.round_recv_off, back, self.round_send_off, Fp, tag="backward")
            for r in range(self.rounds):
                waits[r]()
                self.k.spmm(self.unpack[r], back, dZc[:n_p], accumulate=True)
        if pack is not None:                                # one copy of [dZ | ds2] out of the slab + the K columns of ds1
            if dZc.data_ptr() != pack.data_ptr():
                pack[:, :F + K].copy_(dZc[:n_p, :F + K])
            pack[:, F + K:F + 2 * K].copy_(ds1)
            dZ, ds2 = pack[:, :F], None
        else:
            dZ = dZc[:n_p, :F].clone()
            ds2 = dZc[:n_p, F:F + K].clone()
        if self.mode_id == 1:                               # every Z_j also feeds every row through beta_i
            g = self._allreduce((st.beta.view(n_p, K, 1) * dOut.view(n_p, K, d)).sum(0).reshape(F))
            dZ += g
        return pack if pack is not None else (dZ, ds1, ds2)


class GatAggregate(torch.autograd.Function):
    """out = GAT aggregation of (Z, s1, s2) on ``engine`` with ``state``'s buffers."""

    @staticmethod
    def forward(ctx, engine: GatEngine, state: GatLayerState, Z, s1, s2):
        if state.busy:
            raise RuntimeError("GAT layer state is still owned by a forward whose backward has not run; "
                               "use a fresh state (GatEngine.new_layer_state) for a second forward")
        ctx.engine, ctx.state = engine, state
        state.busy = any(ctx.needs_input_grad)
        return engine.forward(state, Z, s1, s2)           # (Z may be a column slice: it is copied into the panel [Z | s2] anyway)

    @staticmethod
    def backward(ctx, grad_output):
        dZ, ds1, ds2 = ctx.engine.backward(ctx.state, grad_output)
        ctx.state.busy = False
        return None, None, dZ, ds1, ds2


class GatAggregatePacked(torch.autograd.Function):
    """The same aggregation on the PACKED projection ZS = [Z | s2 | s1] (n x (F + 2K)): PGAT computes Z and both attention
    projections as ONE product H . [W^T | W^T a2 | W^T a1] (r05: the two per-head `einsum`s of PGAT.py:141-142 ran as skinny
    batched GEMMs of 1.09 ms each at the benchmark shape).  One gradient comes back, written in place as [dZ | ds2 | ds1]."""

    @staticmethod
    def forward(ctx, engine: GatEngine, state: GatLayerState, ZS):
        if state.busy:
            raise RuntimeError("GAT layer state is still owned by a forward whose backward has not run; "
                               "use a fresh state (GatEngine.new_layer_state) for a second forward")
        K, F = state.heads, state.heads * state.d
        if ZS.dim() != 2 or ZS.shape[1] != F + 2 * K:
            raise ValueError("packed projection must be n x (F + 2K)")
        ctx.engine, ctx.state = engine, state
        state.busy = any(ctx.needs_input_grad)
        if "panel" in __import__("inspect").signature(engine.forward).parameters:
            return engine.forward(state, ZS[:, :F], ZS[:, F + K:F + 2 * K], ZS[:, F:F + K], panel=ZS)
        return engine.forward(state, ZS[:, :F], ZS[:, F + K:F + 2 * K], ZS[:, F:F + K])

    @staticmethod
    def backward(ctx, grad_output):
        st = ctx.state
        K, F = st.heads, st.heads * st.d
        pack = torch.empty((grad_output.shape[0], F + 2 * K), dtype=torch.float32, device=grad_output.device)
        out = ctx.engine.backward(st, grad_output, pack) if _takes_pack(ctx.engine) else None
        if out is None:
            dZ, ds1, ds2 = ctx.engine.backward(st, grad_output)
            pack[:, :F].copy_(dZ); pack[:, F:F + K].copy_(ds2); pack[:, F + K:].copy_(ds1)
        st.busy = False
        return None, None, pack


def _takes_pack(engine) -> bool:
    import inspect
    return "pack" in inspect.signature(engine.backward).parameters
