#!/usr/bin/env python3
"""N1 measurement: MatrixMarket ingest, native (libpgcn_hip.so) vs scipy.io.mmread, on this host."""
import importlib, os, sys, time
import numpy as np, scipy.sparse as sp
from scipy.io import mmread as sp_mmread
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ingest = importlib.import_module("scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd.ingest")
n, nnz = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
path = "/tmp/pgcn_loader_bench.mtx"
rng = np.random.default_rng(0)
r, c, v = rng.integers(1, n + 1, nnz), rng.integers(1, n + 1, nnz), rng.random(nnz)
t = time.time()
with open(path, "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, nnz))
    np.savetxt(f, np.c_[r, c, v], fmt="%d %d %.6e")
print("wrote %.0f MB in %.1f s" % (os.path.getsize(path) / 1e6, time.time() - t))
mb = os.path.getsize(path) / 1e6
for name, fn in (("native(all threads)", lambda: ingest.mmread(path)), ("native(1 thread)", lambda: ingest.mmread(path, 1)), ("scipy.io.mmread", lambda: sp.coo_matrix(sp_mmread(path)))):
    ts = []
    for _ in range(3):
        t = time.time(); A = fn(); ts.append(time.time() - t)
    print("%-22s best %.2f s  %.0f MB/s  %.1f M entries/s  (nnz %d)" % (name, min(ts), mb / min(ts), nnz / min(ts) / 1e6, A.nnz))
os.remove(path)
