#!/usr/bin/env python3
"""The reference's CPU engine ITSELF on a bench workload: /root/reference/Parallel-GCN/main.c, unmodified, built as
oracle/_ref/grbgcn against the GraphBLAS / MPI stand-ins of oracle/shim/ (`make -C oracle ref`),
timed by its own `time : %f secs` line (3 epochs, main.c:229-445).  The stand-in GraphBLAS is a plain single-threaded
CSR implementation -- NOT SuiteSparse -- so this is a floor for what the reference does per rank-thread, recorded next
to baseline B1 (tools/time_reference_b1.py: the reference's GPU/PGCN.py on the CPU) and the OpenMP port in bench.py's
`cpu_baseline`.  Build container only.

usage: python tools/time_reference_grbgcn.py [--workload mid] [--ranks 1] [--out profiles/r03_cpu_reference_grbgcn_mid.json]"""
import argparse, importlib, json, os, re, subprocess, sys, tempfile, time
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mid")
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    synth, io_ = importlib.import_module(PKG + ".synth"), importlib.import_module(PKG + ".pargcn_io")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"], check=True)
    n, _, f, L = synth.SHAPES[a.workload]
    n, row, col, val = synth.make_graph(a.workload, seed=0)
    A = sp.csr_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    pv = synth.random_partvec(n, a.ranks, seed=0).numpy() if a.ranks > 1 else np.zeros(n, np.int64)
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.time()
        io_.write_directory(tmp, A, pv, a.ranks, L, f, value_format="%.9g")
        t_write = time.time() - t0
        t0 = time.time()
        res = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "grbgcn"), "-p", tmp, "-c", os.path.join(tmp, "config"), "-t", "1"],
                             env=dict(os.environ, MPISHIM_NP=str(a.ranks), MPISHIM_SEED="1"), capture_output=True, text=True, timeout=3600)
        wall = time.time() - t0
    assert res.returncode == 0, res.stderr[-2000:]
    secs = float(re.search(r"time : ([0-9.]+) secs", res.stdout).group(1))
    rec = {"baseline": "the reference's Parallel-GCN/main.c (unmodified) on the GraphBLAS / MPI stand-ins of oracle/shim (single-threaded CSR "
                       "GraphBLAS, NOT SuiteSparse)", "workload": a.workload, "n": n, "nnz": int(A.nnz), "f": f, "layers": L,
           "ranks": a.ranks, "threads_per_rank": 1, "host_cpus": os.cpu_count(), "time_3_epochs_s": secs, "ms_per_epoch": 1e3 * secs / 3,
           "edges_per_s": 2 * (L - 1) * A.nnz / (secs / 3), "aggregations_per_epoch": 2 * (L - 1),
           "wall_s_incl_text_parse": wall, "write_directory_s": t_write, "err_lines": re.findall(r"^err:(\S+)$", res.stdout, re.M),
           "note": "main.c runs L - 1 weight layers (config `L n f .. f 2`): 2 (L - 1) aggregations per epoch; its own timer excludes "
                   "the fscanf parse of the data directory"}
    out = a.out or os.path.join(ROOT, "profiles", "r03_cpu_reference_grbgcn_%s.json" % a.workload)
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
