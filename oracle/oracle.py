"""CPU oracle for the GCN aggregation hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the reported CPU
baseline.  The product package never imports it; the product path fails loudly
when its HIP library is missing instead of falling back to anything in here.

Two layers:

* ``lib()``            -- ctypes binding of ``oracle/pgcn_oracle.c`` (plain C
                          restatement of ``Parallel-GCN/main.c:GCN()`` and of the
                          aggregation operator; fp32 storage + accumulation).
* numpy functions      -- independent restatements used to cross-check the C
                          code (float64 "shadow" arithmetic), the host-side
                          helpers of ``GPU/PGCN.py`` (communication maps, the
                          ``run()`` training loop with ReLU / log_softmax /
                          nll_loss / Adam), ``preprocess/GrB-GNN-IDG.py``'s
                          normalisation, and the GAT layer of ``GPU/PGAT.py`` on
                          the stored entries (``gat_*_np``; pinned to outputs and
                          gradients of the reference's own dense layers,
                          ``tests/golden/make_golden_gat.py``).

Every function cites the reference file:line it follows (paths relative to
``/root/reference``).  Parity pinning is described in ``pgcn_oracle.c``'s
header: the aggregation operator and the PGCN.py training loop are pinned by
golden vectors generated from the reference's own ``GPU/PGCN.py``
(``tests/golden/make_golden.py``); the GraphBLAS/MPI training loop of
``Parallel-GCN/main.c`` is pinned by the outputs of that very file, compiled
UNMODIFIED and run in the build container (``make -C oracle ref``:
SuiteSparse:GraphBLAS and MPI are absent and unfetchable, so the calls main.c
makes into them resolve to the minimal stand-ins under ``oracle/shim/`` --
written from the published interfaces, see the header of
``oracle/shim/GraphBLAS.h`` for what that does and does not prove;
``tests/golden/make_pargcn_ref.py`` -> ``tests/golden/pargcn_ref_*``,
``tests/test_reference_grbgcn.py``: the C loop ends on the binary's weights
bit for bit in all ten cases, P = 1, 2, 3, 4, L = 2, 3, 4).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_FAST = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    """Compile oracle/pgcn_oracle.c with gcc (``make -C oracle``)."""
    so = os.path.join(_HERE, "_build", "libpgcn_oracle.so")
    src = os.path.join(_HERE, "pgcn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def _bind(path: str):
    L = ctypes.CDLL(path)
    L.oracle_spmm_csr_f32.argtypes = [ctypes.c_int64, _i64p, _i32p, _f32p, _f32p, ctypes.c_int64,
                                      _f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int]
    L.oracle_spmm_csr_f32.restype = None
    L.oracle_spmm_csr_rows_f32.argtypes = [ctypes.c_int64, _i32p, _i64p, _i32p, _f32p, _f32p,
                                           ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.c_int]
    L.oracle_spmm_csr_rows_f32.restype = None
    L.oracle_gather_rows_f32.argtypes = [_f32p, ctypes.c_int64, _i32p, ctypes.c_int64, _f32p,
                                         ctypes.c_int64, ctypes.c_int32]
    L.oracle_gather_rows_f32.restype = None
    L.oracle_scatter_rows_f32.argtypes = [_f32p, ctypes.c_int64, _i32p, ctypes.c_int64, _f32p,
                                          ctypes.c_int64, ctypes.c_int32, ctypes.c_int]
    L.oracle_scatter_rows_f32.restype = None
    L.oracle_dist_aggregate_f32.argtypes = [ctypes.c_int64, _i64p, _i32p, _f32p, _i32p,
                                            ctypes.c_int32, _f32p, ctypes.c_int64, _f32p,
                                            ctypes.c_int64, ctypes.c_int32]
    L.oracle_dist_aggregate_f32.restype = None
    L.oracle_pargcn_train.argtypes = [ctypes.c_int64, _i64p, _i32p, _f32p, _i32p, ctypes.c_int32,
                                      ctypes.c_int32, _i32p, ctypes.POINTER(_f32p), _f32p, _f32p,
                                      _u8p, ctypes.c_int32, ctypes.c_float, _f32p, _f32p, _i64p]
    L.oracle_pargcn_train.restype = ctypes.c_int
    L.oracle_num_threads.restype = ctypes.c_int
    return L


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "libpgcn_oracle.so")
        if not os.path.exists(so):
            so = build()
        _LIB = _bind(so)
    return _LIB


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def _csr_arrays(A: sp.csr_matrix):
    A = A.tocsr()
    rowptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    col = np.ascontiguousarray(A.indices, dtype=np.int32)
    val = np.ascontiguousarray(A.data, dtype=np.float32)
    return rowptr, col, val


# --------------------------------------------------------------------------
# C-backed operators


def spmm_csr(rowptr: np.ndarray, col: np.ndarray, val: np.ndarray, B: np.ndarray,
             C: Optional[np.ndarray] = None, accumulate: bool = False) -> np.ndarray:
    """C = A.B (or C += A.B), fp32, CSR order.  Parallel-GCN/main.c:271."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    nrows = rowptr.shape[0] - 1
    f = B.shape[1]
    if C is None:
        C = np.zeros((nrows, f), dtype=np.float32)
    assert C.dtype == np.float32 and C.flags.c_contiguous and C.shape == (nrows, f)
    lib().oracle_spmm_csr_f32(nrows, _p(rowptr, _i64p), _p(col, _i32p), _p(val, _f32p),
                              _p(B, _f32p), B.shape[1], _p(C, _f32p), f, f, int(accumulate))
    return C


def spmm(A: sp.spmatrix, B: np.ndarray) -> np.ndarray:
    rowptr, col, val = _csr_arrays(A)
    return spmm_csr(rowptr, col, val, B)


def gather_rows(H: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """GPU/PGCN.py:104 ``H[indices]``."""
    H = np.ascontiguousarray(H, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty((idx.shape[0], H.shape[1]), dtype=np.float32)
    lib().oracle_gather_rows_f32(_p(H, _f32p), H.shape[1], _p(idx, _i32p), idx.shape[0],
                                 _p(out, _f32p), H.shape[1], H.shape[1])
    return out


def scatter_rows(H: np.ndarray, idx: np.ndarray, src: np.ndarray, accumulate: bool) -> None:
    """GPU/PGCN.py:115 ``X[indices] = buf`` / accumulate form of main.c:295,400."""
    assert H.dtype == np.float32 and H.flags.c_contiguous
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    src = np.ascontiguousarray(src, dtype=np.float32)
    lib().oracle_scatter_rows_f32(_p(H, _f32p), H.shape[1], _p(idx, _i32p), idx.shape[0],
                                  _p(src, _f32p), src.shape[1], H.shape[1], int(accumulate))


def dist_aggregate(A: sp.spmatrix, part: Sequence[int], P: int, H: np.ndarray) -> np.ndarray:
    """AH = A.H with the local-then-per-source summation order of main.c:271,295."""
    rowptr, col, val = _csr_arrays(A)
    part = np.ascontiguousarray(part, dtype=np.int32)
    H = np.ascontiguousarray(H, dtype=np.float32)
    n, f = H.shape
    out = np.empty((n, f), dtype=np.float32)
    lib().oracle_dist_aggregate_f32(n, _p(rowptr, _i64p), _p(col, _i32p), _p(val, _f32p),
                                    _p(part, _i32p), P, _p(H, _f32p), f, _p(out, _f32p), f, f)
    return out


def pargcn_train(A: sp.spmatrix, part: Sequence[int], P: int, d: Sequence[int],
                 W: Dict[int, np.ndarray], H0: np.ndarray, Y: np.ndarray, Ymask: np.ndarray,
                 epochs: int = 3, alpha: float = 0.01):
    """Parallel-GCN/main.c:GCN() with P virtual ranks (C, fp32).

    ``d`` = nneurons (d[0]=n); ``W[l]`` (l=1..L-1) is d[l] x d[l+1].  Returns
    (err per epoch, updated W dict, output H_{L-1}, stats[P,2])."""
    rowptr, col, val = _csr_arrays(A)
    part = np.ascontiguousarray(part, dtype=np.int32)
    d = np.ascontiguousarray(d, dtype=np.int32)
    L = d.shape[0] - 1
    n = int(d[0])
    Wc = {l: np.array(W[l], dtype=np.float32, order="C", copy=True) for l in range(1, L)}
    arr = (_f32p * L)()
    for l in range(1, L):
        arr[l] = _p(Wc[l], _f32p)
    H0 = np.ascontiguousarray(H0, dtype=np.float32)
    Y = np.ascontiguousarray(Y, dtype=np.float32)
    Ymask = np.ascontiguousarray(Ymask, dtype=np.uint8)
    err = np.zeros(epochs, dtype=np.float32)
    Hl = np.zeros((n, int(d[L])), dtype=np.float32)
    stats = np.zeros((P, 2), dtype=np.int64)
    rc = lib().oracle_pargcn_train(n, _p(rowptr, _i64p), _p(col, _i32p), _p(val, _f32p),
                                   _p(part, _i32p), P, L, _p(d, _i32p), arr, _p(H0, _f32p),
                                   _p(Y, _f32p), _p(Ymask, _u8p), epochs, alpha,
                                   _p(err, _f32p), _p(Hl, _f32p), _p(stats, _i64p))
    if rc != 0:
        raise ValueError("oracle_pargcn_train rc=%d" % rc)
    return err, Wc, Hl, stats


def delivered_rows(conn: Sequence, part: Sequence[int], P: int) -> np.ndarray:
    """vis[p, j]: rank p ever holds row j of H / G -- its own rows plus the rows its peers' ``conn.q`` files list for
    target p (main.c:526-551 builds the selectors Hsend[target] from those lists, :250 sends exactly those rows,
    :293-295 multiplies by nothing else).  ``conn[q]`` = (``{target: ids}``, nrecvs) as pargcn_io.read_connectivity
    returns it."""
    part = np.asarray(part)
    vis = np.zeros((P, part.shape[0]), dtype=bool)
    for p in range(P):
        vis[p, part == p] = True
    for q in range(P):
        for t, ids in conn[q][0].items():
            vis[t, np.asarray(ids, dtype=np.int64)] = True
    return vis


def drop_undelivered(A: sp.spmatrix, part: Sequence[int], conn: Sequence, P: int) -> Tuple[sp.csr_matrix, int]:
    """The matrix the reference's CPU engine effectively multiplies by, and the number of stored entries it ignores.

    GCN-HP writes the send lists from the ROWS of the sender (GCN-HP/main.cpp:147-176: vertex i goes to the owners
    of the columns of row i), which is what the receivers' rows need only when the pattern is symmetric.  On an
    unsymmetric pattern (HB/gemat11 in tests/golden/pargcn/) rank p never receives some rows its entries refer to;
    main.c then simply has no H row for those columns (H holds owned rows only, Hcap the delivered ones), i.e. the
    entries contribute nothing.  Found when the reference's own main.c was run here (tests/golden/make_pargcn_ref.py):
    with those entries dropped the restated loop ends on the binary's weights bit for bit, without them it is 1e-2
    away.  Rows delivered but referred to by no entry change nothing (they only count in the statistics)."""
    part = np.asarray(part, dtype=np.int64)
    vis = delivered_rows(conn, part, P)
    A = sp.coo_matrix(A)
    keep = vis[part[A.row], A.col]
    out = sp.csr_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=A.shape)
    out.sort_indices()
    return out, int((~keep).sum())


def pargcn_statistics(conn: Sequence, d: Sequence[int], P: int, epochs: int = 3) -> List[int]:
    """The eight numbers of statistics() (main.c:506-524) from the connectivity lists alone: a message to target t
    carries ``len(list) * width`` scalars (the selected rows of a dense H or G, nvals of main.c:253,264), once per
    layer forward (widths d[1..L-1]) and once per layer backward (widths d[L..2]) in each epoch; a receiver counts
    the same numbers on arrival (:287-288)."""
    L = len(d) - 1
    widths = [d[l] for l in range(1, L)] + [d[l + 1] for l in range(L - 1, 0, -1)]
    send_vol, recv_vol = np.zeros(P, np.int64), np.zeros(P, np.int64)
    send_msg, recv_msg = np.zeros(P, np.int64), np.zeros(P, np.int64)
    for q in range(P):
        for t, ids in conn[q][0].items():
            rows = len(ids)
            send_vol[q] += rows * sum(widths) * epochs
            recv_vol[t] += rows * sum(widths) * epochs
            send_msg[q] += len(widths) * epochs
            recv_msg[t] += len(widths) * epochs
    return [int(send_vol.sum()), int(send_vol.sum() // P), int(send_vol.max()), int(recv_vol.max()),
            int(send_msg.sum()), int(send_msg.sum() // P), int(send_msg.max()), int(recv_msg.max())]


# --------------------------------------------------------------------------
# numpy restatements (independent of the C code)


def normalize_adjacency(A: sp.spmatrix) -> sp.csr_matrix:
    """preprocess/GrB-GNN-IDG.py:45-68: A_hat = Dr^-1/2 (A - diag + I) Dc^-1/2."""
    A = sp.coo_matrix(A, dtype=np.float64)
    keep = A.row != A.col  # :47-50 zero the diagonal then eliminate
    A = sp.coo_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=A.shape).tocsr()
    n = A.shape[0]
    A = A + sp.identity(n, format="csr")  # :53-54
    col_sum = 1.0 / np.sqrt(np.asarray(A.sum(axis=0)).reshape(-1))  # :56-59
    row_sum = 1.0 / np.sqrt(np.asarray(A.sum(axis=1)).reshape(-1))  # :63-66
    A = sp.diags(row_sum) @ A @ sp.diags(col_sum)  # :70
    A = A.tocsr()
    A.sort_indices()
    return A.astype(np.float32)


def communication_maps(A: sp.spmatrix, partvec: Sequence[int], rank: int, size: int
                       ) -> Tuple[Dict[int, np.ndarray], Dict[int, np.ndarray]]:
    """GPU/PGCN.py:37-51 compute_communication_maps, vectorised.

    recv_map[q] = sorted unique columns owned by q that appear in my rows;
    send_map[q] = sorted unique columns owned by me that appear in q's rows;
    the own rank is popped (:49-50).  Stored explicit zeros count as entries,
    exactly like the reference's loop over ``A.nnz``."""
    A = sp.coo_matrix(A)
    pv = np.asarray(partvec, dtype=np.int64)
    pr, pc = pv[A.row], pv[A.col]
    send_map, recv_map = {}, {}
    for q in range(size):
        if q == rank:
            continue
        recv_map[q] = np.unique(A.col[(pr == rank) & (pc == q)]).astype(np.int64)
        send_map[q] = np.unique(A.col[(pc == rank) & (pr == q)]).astype(np.int64)
    return send_map, recv_map


def dist_aggregate_messages(A: sp.spmatrix, part: Sequence[int], P: int, H: np.ndarray,
                            dtype=np.float64) -> Tuple[np.ndarray, np.ndarray]:
    """Explicit message-passing restatement of main.c:238-299 with P rank objects.

    Every rank holds only its owned rows, packs ``H[S_pq]`` for each target
    (main.c:245-267), computes the local product (271) and accumulates one
    product per received message (275-299).  Returns (AH, rows_sent[p, q])."""
    A = sp.csr_matrix(A)
    part = np.asarray(part)
    n, f = H.shape
    Hd = H.astype(dtype)
    out = np.zeros((n, f), dtype=dtype)
    rows_sent = np.zeros((P, P), dtype=np.int64)
    owned = [np.nonzero(part == p)[0] for p in range(P)]
    maps = [communication_maps(A, part, p, P) for p in range(P)]
    # "network": mailbox[(src, dst)] = (global ids, rows)
    mailbox = {}
    for p in range(P):
        send_map, _ = maps[p]
        for q, ids in send_map.items():
            if ids.size:
                mailbox[(p, q)] = (ids, Hd[ids].copy())
                rows_sent[p, q] = ids.size
    for p in range(P):
        Ap = A[owned[p]].astype(dtype)
        # local piece: a rank's H has no rows outside its part
        Hloc = np.zeros((n, f), dtype=dtype)
        Hloc[owned[p]] = Hd[owned[p]]
        acc = Ap @ Hloc
        for q in range(P):
            if (q, p) not in mailbox:
                continue
            ids, rows = mailbox[(q, p)]
            Hcap = np.zeros((n, f), dtype=dtype)
            Hcap[ids] = rows  # GrB_Matrix_build of the received tuples, main.c:293
            acc = acc + Ap @ Hcap  # main.c:295
        out[owned[p]] = acc
    return out, rows_sent


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def pargcn_train_np(A: sp.spmatrix, d: Sequence[int], W: Dict[int, np.ndarray], H0: np.ndarray,
                    Y: np.ndarray, Ymask: np.ndarray, epochs: int = 3, alpha: float = 0.01,
                    dtype=np.float64):
    """float64 shadow of Parallel-GCN/main.c:GCN() (single address space; the
    partition only affects fp32 summation order, which float64 arbitrates)."""
    A = sp.csr_matrix(A).astype(dtype)
    L = len(d) - 1
    n = d[0]
    W = {l: np.array(W[l], dtype=dtype) for l in range(1, L)}
    H = {0: H0.astype(dtype)}
    Z = {}
    errs = []
    Ym = Ymask.astype(bool)
    Yd = Y.astype(dtype)
    for _ in range(epochs):
        for l in range(1, L):  # main.c:233
            AH = A @ H[l - 1]
            Z[l] = AH @ W[l]  # :303
            H[l] = _sigmoid(Z[l])  # :308
        Pm = H[L - 1]
        T = np.where(Ym, -1.0 * Yd * np.log(np.where(Ym, Pm, 1.0)), Pm)  # :318 (union)
        errs.append(T.sum())  # :320
        D = np.where(Ym, Pm - Yd, Pm) / (Pm * (1 - Pm))  # :325-328
        s = _sigmoid(Z[L - 1])
        G = {L - 1: D * (s * (1 - s)) / n}  # :330-335
        for l in range(L - 1, 0, -1):  # :338
            AG = A @ G[l]  # :376 (A, not A^T)
            if l != 1:
                s = _sigmoid(Z[l - 1])
                G[l - 1] = (AG @ W[l].T) * (s * (1 - s))  # :407-410
            dW = H[l - 1].T @ AG  # :417
            W[l] = W[l] - alpha * dW  # :430
    return np.array(errs), W, H[L - 1]


# ---- GPU/PGCN.py training loop (ReLU / log_softmax / nll / Adam) ---------


def _log_softmax(x):
    m = x.max(axis=1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


class _Adam:
    """torch.optim.Adam defaults (lr given, betas 0.9/0.999, eps 1e-8), PGCN.py:200."""

    def __init__(self, params: List[np.ndarray], lr: float):
        self.p, self.lr = params, lr
        self.m = [np.zeros_like(p) for p in params]
        self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        for p, g, m, v in zip(self.p, grads, self.m, self.v):
            m *= b1
            m += (1 - b1) * g
            v *= b2
            v += (1 - b2) * g * g
            bc1 = 1 - b1 ** self.t
            bc2 = 1 - b2 ** self.t
            denom = np.sqrt(v) / np.sqrt(bc2) + eps
            p -= (self.lr / bc1) * m / denom


def pgcn_train_np(A: sp.spmatrix, part: Sequence[int], P: int, weights: List[np.ndarray],
                  H0: np.ndarray, labels: np.ndarray, epochs: int = 5, lr: float = 1e-3,
                  dtype=np.float64, schedule=None):
    """GPU/PGCN.py:run() 194-226 with exact aggregation (no Q1-Q3 quirks).

    ``weights[l]`` is the ``nn.Linear`` weight (out x in) of layer l.  Returns
    (rank-0 loss per epoch, final weights).  Loss semantics follow the
    reference literally (Q4): every rank averages nll over ALL n rows of an
    n x f logits matrix whose non-owned rows are zero (PGCN.py:213-215), the
    per-rank gradients are summed and divided by P (:150-154).

    ``schedule`` (optional): one adjacency matrix per optimiser step instead of ``epochs`` steps on ``A`` -- the
    mini-batch driver (GPU/PGCN-Mini-batch.py:252-296: every batch is a step on its induced sub-adjacency).  The
    returned losses are then per STEP, summed over the ranks like :294 (mean nll + (P - 1) log f)."""
    part = np.asarray(part)
    if schedule is not None:
        n, f = H0.shape
        Ws = [np.array(w, dtype=dtype) for w in weights]
        opt = _Adam(Ws, lr)
        H0 = H0.astype(dtype)
        onehot = np.zeros((n, f), dtype=dtype)
        onehot[np.arange(n), labels] = 1
        losses = []
        for At in schedule:
            At = sp.csr_matrix(At).astype(dtype)
            acts, pre, agg = [H0], [], []
            for W in Ws:
                AH = At @ acts[-1]
                agg.append(AH)
                pre.append(AH @ W.T)
                acts.append(np.maximum(pre[-1], 0))
            logp = _log_softmax(acts[-1])
            losses.append(-(logp * onehot).sum() / n + (P - 1) * np.log(f))
            g = (np.exp(logp) - onehot) / n / P
            grads = [None] * len(Ws)
            for l in range(len(Ws) - 1, -1, -1):
                g = g * (pre[l] > 0)
                grads[l] = g.T @ agg[l]
                g = At.T @ (g @ Ws[l])
            opt.step(grads)
        return np.array(losses), Ws
    A = sp.csr_matrix(A).astype(dtype)
    n, f = H0.shape
    Ws = [np.array(w, dtype=dtype) for w in weights]
    opt = _Adam(Ws, lr)
    H0 = H0.astype(dtype)
    onehot = np.zeros((n, f), dtype=dtype)
    onehot[np.arange(n), labels] = 1
    losses = []
    own0 = part == 0
    for _ in range(epochs):
        acts, pre, agg = [H0], [], []
        for W in Ws:  # PGCN.forward :144-148
            AH = A @ acts[-1]
            Zl = AH @ W.T
            agg.append(AH)
            pre.append(Zl)
            acts.append(np.maximum(Zl, 0))
        logits = acts[-1]
        logp = _log_softmax(logits)
        nll = -(logp * onehot).sum(axis=1)
        # rank 0's printed loss: owned rows + (n - n_0) rows of zero logits (nll = log f)
        losses.append((nll[own0].sum() + (n - own0.sum()) * np.log(f)) / n)
        # sum over ranks of d(local mean over n)/dW, then / P (average_gradients)
        g = (np.exp(logp) - onehot) / n / P
        grads = [None] * len(Ws)
        for l in range(len(Ws) - 1, -1, -1):
            g = g * (pre[l] > 0)
            grads[l] = g.T @ agg[l]
            g = A.T @ (g @ Ws[l])  # PSpMM.backward :130-134 (A^T)
        opt.step(grads)
    return np.array(losses), Ws


def pgcn_epochs_f32(csr, csr_t, weights: List[np.ndarray], H0: np.ndarray, labels: np.ndarray, epochs: int,
                    lr: float = 1e-3):
    """The epoch of GPU/PGCN.py:run() 212-220 at P = 1 in fp32 on the host cores, for bench.py's `cpu_baseline` leg:
    per layer  AH = A.H (OpenMP CSR SpMM of pgcn_oracle.c), Z = AH.W^T, H = relu(Z)  (PGCN.forward :144-148);
    log_softmax + nll_loss (:213-214); the backward pass with A^T (PSpMM.backward :130-134), the three GEMMs of every
    layer and torch.optim.Adam's defaults (:200).  The same arithmetic as pgcn_train_np (the float64 shadow the GPU
    tests use), organised for speed: BLAS GEMMs (numpy), in-place element-wise passes.  ``csr`` / ``csr_t``:
    (rowptr int64, col int32, val fp32) of A and A^T.  Returns (losses, seconds per epoch)."""
    import time
    n, f = H0.shape
    Ws = [np.array(w, dtype=np.float32) for w in weights]
    opt = _Adam(Ws, lr)
    H0 = np.ascontiguousarray(H0, dtype=np.float32)
    rows = np.arange(n)
    losses, secs = [], []
    for _ in range(epochs):
        t0 = time.time()
        acts, agg = [H0], []
        for W in Ws:
            AH = spmm_csr(*csr, acts[-1])
            agg.append(AH)
            Z = AH @ W.T
            np.maximum(Z, 0, out=Z)
            acts.append(Z)                                   # relu(Z); Z > 0 <=> relu(Z) > 0 for the backward mask
        logits = acts[-1]
        m = logits.max(axis=1, keepdims=True)
        z = logits - m
        np.exp(z, out=z)
        ssum = z.sum(axis=1, keepdims=True)
        losses.append(float((np.log(ssum[:, 0]) + m[:, 0] - logits[rows, labels]).mean()))
        g = z / ssum                                         # softmax
        g[rows, labels] -= 1
        g *= np.float32(1.0 / n)
        grads = [None] * len(Ws)
        for l in range(len(Ws) - 1, -1, -1):
            g *= acts[l + 1] > 0
            grads[l] = g.T @ agg[l]
            g = spmm_csr(*csr_t, g @ Ws[l])
        opt.step(grads)
        secs.append(time.time() - t0)
    return losses, secs


# ---------------------------------------------------------------------------------------------
# GAT path (GPU/PGAT.py) -- numpy restatement, vectorised per row segment (no C code: the dense
# reference is a few matmuls; this sparse restatement is pinned to it by tests/golden/ref_gat_*).
#
#   mode "reference": the literal arithmetic of PGAT.forward (PGAT.py:138-151):
#       Z = H W^T ; z1 = Z a[:F] ; z2 = Z a[F:]                       (:140-142)
#       att = z1 + z2^T ; att = where(A > 0, att, 0)                  (:144-146)   <- non-edges get logit 0, not -inf
#       att = softmax(att, dim=1) over ALL n columns ; out = att Z    (:147-149)
#     restated sparsely: with m_i = max(0, max_j e_ij), D_i = sum_edges exp(e_ij-m_i) + (n-deg_i) exp(-m_i),
#       out_i = sum_edges (p_ij - b_i) Z_j + b_i sum_all Z_j,   p_ij = exp(e_ij-m_i)/D_i,  b_i = exp(-m_i)/D_i
#   mode "standard": e_ij = LeakyReLU(z1_i + z2_j), softmax over the neighbours only, K heads
#     (Velickovic et al.; the semantics BASELINE config 5 names: edge-softmax + weighted SpMM).
#
# Layout: Z is (n_cols, K*d) with head k in columns [k*d, (k+1)*d); s1 (n_rows, K), s2 (n_cols, K).


def _segment_ids(rowptr: np.ndarray) -> np.ndarray:
    return np.repeat(np.arange(rowptr.shape[0] - 1), np.diff(rowptr))


def gat_scores_np(A: sp.csr_matrix, s1: np.ndarray, s2: np.ndarray, mode: str, slope: float, n_global: int):
    """Edge weights of one layer.  Returns (alpha [nnz,K] in CSR order, beta [n_rows,K], p [nnz,K]).
    standard: alpha = softmax over the row's entries, beta = 0, p = alpha.
    reference: alpha = p - beta (see header)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    rowptr, col = A.indptr.astype(np.int64), A.indices.astype(np.int64)
    nr, K = A.shape[0], s1.shape[1]
    dt = s1.dtype
    seg = _segment_ids(rowptr)
    raw = s1[seg] + s2[col]                                             # PGAT.py:144
    nonempty = np.diff(rowptr) > 0
    starts = rowptr[:-1][nonempty]
    if mode == "standard":
        e = np.where(raw > 0, raw, raw * dt.type(slope))
        m = np.full((nr, K), -np.inf, dtype=dt)
        if e.shape[0]:
            m[nonempty] = np.maximum.reduceat(e, starts, axis=0)
        w = np.exp(e - m[seg])
        D = np.zeros((nr, K), dtype=dt)
        if e.shape[0]:
            D[nonempty] = np.add.reduceat(w, starts, axis=0)
        alpha = w / D[seg]
        return alpha.astype(dt), np.zeros((nr, K), dtype=dt), alpha.astype(dt)
    if mode != "reference":
        raise ValueError(mode)
    m = np.zeros((nr, K), dtype=dt)
    if raw.shape[0]:
        m[nonempty] = np.maximum(np.maximum.reduceat(raw, starts, axis=0), 0)
    em = np.exp(-m)
    w = np.exp(raw - m[seg])
    deg = np.diff(rowptr).astype(dt)[:, None]
    D = (dt.type(n_global) - deg) * em
    if raw.shape[0]:
        D[nonempty] += np.add.reduceat(w, starts, axis=0)
    p = w / D[seg]
    beta = em / D
    return (p - beta[seg]).astype(dt), beta.astype(dt), p.astype(dt)


def gat_aggregate_np(A: sp.csr_matrix, Z: np.ndarray, s1: np.ndarray, s2: np.ndarray, mode: str = "standard",
                     slope: float = 0.2, n_global: Optional[int] = None, Zsum: Optional[np.ndarray] = None):
    """out[i,k,:] = sum_j alpha_ijk Z[j,k,:] (+ beta_ik Zsum[k,:] in reference mode).  A is the
    pattern (n_rows x n_cols; reference mode: the entries with A_ij > 0, PGAT.py:146)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    nr, nc = A.shape
    K = s1.shape[1]
    d = Z.shape[1] // K
    n_global = nc if n_global is None else n_global
    alpha, beta, _ = gat_scores_np(A, s1, s2, mode, slope, n_global)
    out = np.zeros((nr, K * d), dtype=Z.dtype)
    for k in range(K):
        Ak = sp.csr_matrix((alpha[:, k], A.indices, A.indptr), shape=A.shape)
        out[:, k * d:(k + 1) * d] = Ak @ Z[:, k * d:(k + 1) * d]
    if mode == "reference":
        if Zsum is None:
            Zsum = Z.sum(axis=0, dtype=Z.dtype)
        out += (beta[:, :, None] * Zsum.reshape(K, d)[None]).reshape(nr, K * d)
    return out


def gat_aggregate_backward_np(A: sp.csr_matrix, Z: np.ndarray, s1: np.ndarray, s2: np.ndarray, dOut: np.ndarray,
                              mode: str = "standard", slope: float = 0.2, n_global: Optional[int] = None,
                              Zsum: Optional[np.ndarray] = None, g: Optional[np.ndarray] = None):
    """Gradients of gat_aggregate_np w.r.t. (Z, s1, s2) given dOut.  ``g`` (reference mode) is
    sum_i beta_i dOut_i over ALL rows of the graph (defaults to the rows of A), added to every
    row of dZ that this call owns -- the caller decides which rows those are: here ALL n_cols."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    nr, nc = A.shape
    K = s1.shape[1]
    d = Z.shape[1] // K
    n_global = nc if n_global is None else n_global
    rowptr, col = A.indptr.astype(np.int64), A.indices.astype(np.int64)
    seg = _segment_ids(rowptr)
    alpha, beta, p = gat_scores_np(A, s1, s2, mode, slope, n_global)
    out = gat_aggregate_np(A, Z, s1, s2, mode, slope, n_global, Zsum)
    t = (dOut.reshape(nr, K, d) * out.reshape(nr, K, d)).sum(-1)                  # <dOut_i, out_i> per head
    dp = (dOut.reshape(nr, K, d)[seg] * Z.reshape(nc, K, d)[col]).sum(-1)         # <dOut_i, Z_j>
    de = p * (dp - t[seg])
    if mode == "standard":
        raw = s1[seg] + s2[col]
        de = de * np.where(raw > 0, 1.0, slope).astype(Z.dtype)
    ds1 = np.zeros((nr, K), dtype=Z.dtype)
    ds2 = np.zeros((nc, K), dtype=Z.dtype)
    np.add.at(ds1, seg, de)
    np.add.at(ds2, col, de)
    dZ = np.zeros_like(Z)
    for k in range(K):
        Ak = sp.csr_matrix((alpha[:, k], A.indices, A.indptr), shape=A.shape)
        dZ[:, k * d:(k + 1) * d] = Ak.T @ dOut[:, k * d:(k + 1) * d]
    if mode == "reference":
        if g is None:
            g = (beta[:, :, None] * dOut.reshape(nr, K, d)).sum(0).reshape(K * d)
        dZ += g[None, :]
    return dZ, ds1, ds2


def gat_layer_np(A: sp.csr_matrix, H: np.ndarray, W: np.ndarray, a: np.ndarray, heads: int = 1,
                 mode: str = "standard", slope: float = 0.2):
    """One PGAT layer on one process (PGAT.py:138-151): W is (F, f_in) like nn.Linear.weight,
    a is (2*d, heads) -- column k holds [a1_k ; a2_k] (the reference's (2F, 1) for one head)."""
    Z = H @ W.T
    n, F = Z.shape
    d = F // heads
    Zh = Z.reshape(n, heads, d)
    s1 = np.einsum("nkd,dk->nk", Zh, a[:d])
    s2 = np.einsum("nkd,dk->nk", Zh, a[d:])
    return gat_aggregate_np(A, Z, s1.astype(Z.dtype), s2.astype(Z.dtype), mode, slope, n), Z, s1, s2


def gat_dense_reference_np(Adense: np.ndarray, H: np.ndarray, W: np.ndarray, a: np.ndarray) -> np.ndarray:
    """The reference's dense arithmetic, literally (PGAT.py:140-149), single head -- used to check
    the sparse restatement above independently of the golden files."""
    Z = H @ W.T
    F = Z.shape[1]
    att = Z @ a[:F] + (Z @ a[F:]).T
    att = np.where(Adense > 0, att, 0.0)
    att = att - att.max(axis=1, keepdims=True)
    e = np.exp(att)
    return (e / e.sum(axis=1, keepdims=True)) @ Z
