"""On-disk inputs of the reference's CPU engine.

Readers for the per-rank files GCN-HP writes and Parallel-GCN/main.c consumes, so the SAME
directories feed the oracle and the HIP engine:

  config   "L n d1 .. d(L-1) 2"                          main.c:687-714  (GCN-HP/main.cpp:117-131)
  A.k Y.k  "nvtx nnz" then "i j value" (0-based, global)   main.c:609-648  (GCN-HP/main.cpp:213-249, %.2f)
  H.k      "nrows" then one owned global row id per line  main.c:650-685  (GCN-HP/main.cpp:251-282)
  conn.k   "ntargets nsources" then "target count ids.."  main.c:526-551  (GCN-HP/main.cpp:178-191)
  buff.k   "ntargets (target rows)*" / "nsources (source rows)*"   main.c:456-504

Host-side text parsing: plumbing, no GPU."""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import scipy.sparse as sp


def read_config(path: str) -> Tuple[int, List[int]]:
    """main.c:687-714: nlayers then nlayers+1 widths (d[0] = number of vertices)."""
    tok = open(path).read().split()
    L = int(tok[0])
    d = [int(t) for t in tok[1:L + 2]]
    if len(d) != L + 1:
        raise ValueError("config %s: expected %d widths, found %d" % (path, L + 1, len(d)))
    return L, d


def read_matrix(path: str) -> Tuple[int, np.ndarray, np.ndarray, np.ndarray]:
    """main.c:609-648: header "nvtx nedges", then nedges triples "i j x"."""
    with open(path) as f:
        nvtx, nedges = (int(t) for t in f.readline().split())
        a = np.loadtxt(f, dtype=np.float64, ndmin=2) if nedges else np.zeros((0, 3))
    if a.shape[0] != nedges:
        raise ValueError("%s: header says %d entries, file holds %d" % (path, nedges, a.shape[0]))
    return nvtx, a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.float32)


def read_rows(path: str) -> np.ndarray:
    """main.c:650-685 (H.k): count, then the owned global row ids (features are all ones)."""
    tok = open(path).read().split()
    n = int(tok[0])
    ids = np.asarray(tok[1:1 + n], dtype=np.int64)
    if ids.size != n:
        raise ValueError("%s: header says %d rows, file holds %d" % (path, n, ids.size))
    return ids


def read_connectivity(path: str) -> Tuple[Dict[int, np.ndarray], int]:
    """main.c:526-551: {target rank: global ids of my rows it needs}, number of sources."""
    tok = open(path).read().split()
    ntargets, nrecvs = int(tok[0]), int(tok[1])
    out, p = {}, 2
    for _ in range(ntargets):
        target, cnt = int(tok[p]), int(tok[p + 1])
        out[target] = np.asarray(tok[p + 2:p + 2 + cnt], dtype=np.int64)
        p += 2 + cnt
    return out, nrecvs


def read_buffer_sizes(path: str) -> Tuple[Dict[int, int], Dict[int, int]]:
    """main.c:456-504: rows to send per target, rows to receive per source."""
    lines = open(path).read().split("\n")
    def parse(line):
        tok = line.split()
        k = int(tok[0]) if tok else 0
        return {int(tok[1 + 2 * i]): int(tok[2 + 2 * i]) for i in range(k)}
    return parse(lines[0]), parse(lines[1] if len(lines) > 1 else "")


def count_parts(directory: str) -> int:
    k = 0
    while os.path.exists(os.path.join(directory, "A.%d" % k)):
        k += 1
    return k


def load_directory(directory: str, config_path: str = None):
    """Assemble the global problem from all per-rank files: (L, d, A coo, partvec, Y dense,
    Ymask, conn[k] dicts, buff[k] pairs)."""
    L, d = read_config(config_path or os.path.join(directory, "config"))
    n = d[0]
    k = count_parts(directory)
    if k == 0:
        raise FileNotFoundError("no A.0 in %s" % directory)
    rows, cols, vals = [], [], []
    part = np.full(n, -1, dtype=np.int64)
    Y = np.zeros((n, d[L]), dtype=np.float32)
    Ymask = np.zeros((n, d[L]), dtype=np.uint8)
    conn, buff = [], []
    for p in range(k):
        nv, r, c, v = read_matrix(os.path.join(directory, "A.%d" % p))
        if nv != n:
            raise ValueError("A.%d: %d vertices, config says %d" % (p, nv, n))
        rows.append(r); cols.append(c); vals.append(v)
        own = read_rows(os.path.join(directory, "H.%d" % p))
        if (part[own] != -1).any():
            raise ValueError("row owned twice (H.%d)" % p)
        part[own] = p
        _, yr, yc, yv = read_matrix(os.path.join(directory, "Y.%d" % p))
        Y[yr, yc] = yv
        Ymask[yr, yc] = 1
        conn.append(read_connectivity(os.path.join(directory, "conn.%d" % p)))
        buff.append(read_buffer_sizes(os.path.join(directory, "buff.%d" % p)))
    if (part < 0).any():
        raise ValueError("%d vertices are owned by no part" % int((part < 0).sum()))
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    if not (part[A.row] == np.repeat(np.arange(k), [len(r) for r in rows])).all():
        raise ValueError("an A.k file holds rows of another part")
    return {"L": L, "d": d, "A": A, "part": part, "Y": Y, "Ymask": Ymask, "conn": conn, "buff": buff, "k": k}


def delivered_rows(conn, part, k: int) -> np.ndarray:
    """vis[p, j]: rank p of the reference engine ever holds row j of H / G -- the rows it owns plus the rows its
    peers' conn files list for target p (main.c:526-551 builds the row selectors from those lists, :250 sends exactly
    those rows, :293-295 multiplies by nothing else).  ``conn`` as load_directory returns it."""
    part = np.asarray(part)
    vis = np.zeros((k, part.shape[0]), dtype=bool)
    for p in range(k):
        vis[p, part == p] = True
    for q in range(k):
        for t, ids in conn[q][0].items():
            vis[t, np.asarray(ids, dtype=np.int64)] = True
    return vis


# --------------------------------------------------------------------------------------------
# Writers: the same files the reference's tools write, so that directories produced
# here feed the reference's CPU engine and vice versa.


def write_partvec(path: str, partvec) -> None:
    """GPU/hypergraph/main.cpp:51-63 (`print_partvec`), GPU/graph/main.cpp:53-65: ONE line, every part id followed
    by a space, then a newline.  File name convention: <matrix file name>.<k>.{hp,gp,rp}."""
    pv = np.asarray(partvec, dtype=np.int64)
    with open(path, "w") as f:
        f.write("".join("%d " % p for p in pv.tolist()))
        f.write("\n")


def write_config(path: str, L: int, n: int, f: int, nout: int = 2) -> None:
    """GCN-HP/main.cpp:117-131 (`print_config`): "L n f .. f nout " (L-1 times f), newline."""
    with open(path, "w") as fh:
        fh.write("%d %d " % (L, n) + "%d " % f * (L - 1) + "%d \n" % nout)


def _write_triples(path: str, n: int, rows, cols, vals, value_format: str) -> None:
    with open(path, "w") as fh:
        fh.write("%d %d\n" % (n, len(rows)))
        fmt = "%d %d " + value_format + "\n"
        fh.write("".join(fmt % (int(i), int(j), float(x)) for i, j, x in zip(rows, cols, vals)))


def communication_lists(A: sp.spmatrix, partvec, k: int):
    """GCN-HP/main.cpp:147-176: Hsend[source][target] = ascending ids of the rows i of `source` that hold an entry
    whose COLUMN belongs to `target` (the reference derives who needs H[i] from row i itself: valid for the
    symmetric matrices `preprocess` writes); Hrecv[target][source] is the same list seen from the receiver."""
    A = sp.csr_matrix(A)
    part = np.asarray(partvec, dtype=np.int64)
    coo = A.tocoo()
    src, tgt = part[coo.row], part[coo.col]
    cut = src != tgt
    key = np.unique((src[cut] * k + tgt[cut]) * A.shape[0] + coo.row[cut])
    pair, ids = key // A.shape[0], key % A.shape[0]
    send = [dict() for _ in range(k)]
    recv = [dict() for _ in range(k)]
    for pr in np.unique(pair):
        s, t = int(pr // k), int(pr % k)
        lst = ids[pair == pr]
        send[s][t] = lst
        recv[t][s] = lst
    return send, recv


def write_directory(directory: str, A: sp.spmatrix, partvec, k: int, L: int, f: int, Y: sp.spmatrix = None,
                    nout: int = 2, value_format: str = "%.2f") -> None:
    """Everything GCN-HP's main() leaves in its output directory (GCN-HP/main.cpp:103-110):
    A.p / Y.p (print_parts, :213-249: header "n nnz_p", then "i j %.2f" for the rows of part p), H.p (print_parts2,
    :251-282: number of owned rows, then their ids), conn.p / buff.p (print_connectivity, :147-211) and config.
    ``Y`` defaults to the reference preprocess' label matrix (column 1 set for every vertex,
    preprocess/GrB-GNN-IDG.py:76-78).  ``value_format="%.2f"`` is what the reference prints (it rounds the
    normalised values to two decimals); pass "%.9g" for a lossless directory."""
    os.makedirs(directory, exist_ok=True)
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    part = np.asarray(partvec, dtype=np.int64)
    if part.shape[0] != n or part.min() < 0 or part.max() >= k:
        raise ValueError("part vector does not fit the matrix / the number of parts")
    if Y is None:
        Y = sp.csr_matrix((np.ones(n, dtype=np.float32), (np.arange(n), np.ones(n, dtype=np.int64))), shape=(n, nout))
    Y = sp.csr_matrix(Y)
    Y.sort_indices()
    write_config(os.path.join(directory, "config"), L, n, f, nout)
    send, recv = communication_lists(A, part, k)
    for p in range(k):
        own = np.nonzero(part == p)[0]
        for name, M in (("A", A), ("Y", Y)):
            sub = M[own].tocoo()
            _write_triples(os.path.join(directory, "%s.%d" % (name, p)), n, own[sub.row], sub.col, sub.data, value_format)
        with open(os.path.join(directory, "H.%d" % p), "w") as fh:
            fh.write("%d\n" % own.size + "".join("%d\n" % i for i in own.tolist()))
        with open(os.path.join(directory, "conn.%d" % p), "w") as fh:
            fh.write("%d %d\n" % (len(send[p]), len(recv[p])))
            for t in sorted(send[p]):
                fh.write("%d %d " % (t, send[p][t].size) + "".join("%d " % i for i in send[p][t].tolist()) + "\n")
        with open(os.path.join(directory, "buff.%d" % p), "w") as fh:
            fh.write("%d " % len(send[p]) + "".join("%d %d " % (t, send[p][t].size) for t in sorted(send[p])))
            fh.write("\n%d " % len(recv[p]) + "".join("%d %d " % (s, recv[p][s].size) for s in sorted(recv[p])))
