#!/usr/bin/env python3
"""Golden outputs of the reference's CPU engine ITSELF: /root/reference/Parallel-GCN/main.c, compiled unmodified
(oracle/Makefile target `ref` -> oracle/_ref/grbgcn; its two absent dependencies, SuiteSparse:GraphBLAS and MPI, are
replaced by the minimal stand-ins under oracle/shim/, see the header of oracle/shim/GraphBLAS.h) and RUN here on data
directories in the reference's on-disk format.  Build container only (needs /root/reference); what it writes --
tests/golden/pargcn_ref_<case>.{json,npz} -- travels.  Usage:  python tests/golden/make_pargcn_ref.py

Per case the fixture holds what the binary printed (the `err:` line of each of its 3 epochs, main.c:323, and the 8
communication statistics, main.c:519-522), the weights it ended with (all ranks agree bit for bit; they leave the
process through the stand-in's GRBSHIM_DUMP hook because main.c prints none), and the weights it started from: main.c
draws them from rand() after srand(time(NULL)) (main.c:555-573); the MPI stand-in pins time() to MPISHIM_SEED and
`pargcn.init_weights(d, seed, "glibc")` restates the draw -- that the restated start is the binary's start is what the
first `err:` line checks.  tests/test_reference_grbgcn.py holds the oracle (and, on the GPU, the HIP engine) to these.

Cases: the two karate directories and the gemat11 directory that the reference's own preprocess + GCN-HP tools wrote
(tests/golden/pargcn/, values rounded to two decimals by those tools), plus three directories written by
pargcn_io.write_directory (byte-identical to the tools' output on the same input, tests/test_formats_order.py) to
cover one rank, two layers (no hidden gradient, main.c:406), four layers and lossless values, and two on the Cora
shape of BASELINE.json configs[0] (the reference's own CPU-runnable case)."""
import json
import os
import re
import shutil
import struct
import subprocess
import sys
import tarfile
import tempfile

import numpy as np
import scipy.sparse as sp
from scipy.io import mmread

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg, read_partvec  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "grbgcn")

# name -> how the data directory comes to be.  "dir"/"tar": committed output of the reference's tools;
# "write": pargcn_io.write_directory(mtx, part vector, k, L, f, value format) -- tests rebuild it the same way.
CASES = {
    "karate_k2": {"dir": "pargcn/karate_k2", "P": 2, "seed": 7},
    "karate_k3": {"dir": "pargcn/karate_k3", "P": 3, "seed": 11},
    "gemat11p_k3": {"tar": "pargcn/gemat11p_k3.tar.gz", "P": 3, "seed": 5},
    # layer widths that differ from layer to layer (GCN-HP only writes "f .. f"; main.c:687-714 reads any list)
    "karate_k3_widths": {"dir": "pargcn/karate_k3", "config": [34, 12, 20, 6, 2], "P": 3, "seed": 17},
    "karate_k1_l2": {"write": {"mtx": "karate.A.mtx", "partvec": "karate.mtx.1.rp", "k": 1, "L": 2, "f": 8,
                               "value_format": "%.9g"}, "P": 1, "seed": 3},
    "gemat11p_k2_l4": {"write": {"mtx": "gemat11p.A.mtx", "partvec": "gemat11.mtx.2.rp", "k": 2, "L": 4, "f": 8,
                                 "value_format": "%.9g"}, "P": 2, "seed": 2024},
    # BASELINE.json configs[0]: the Cora shape (2 708 vertices, 10 556 directed edges), 2 layers, f = 16, on the reference's
    # CPU path with one rank -- and over two ranks under a random part vector.  Synthetic stand-in (no data sets here):
    # synth.make_graph("cora") draws from the portable stream, so the test rebuilds the same directory anywhere.
    "cora_k1": {"synth": {"workload": "cora", "seed": 0, "k": 1, "L": 2, "f": 16}, "P": 1, "seed": 42},
    "cora_k2": {"synth": {"workload": "cora", "seed": 0, "k": 2, "L": 2, "f": 16}, "P": 2, "seed": 43},
    # four ranks, a power-law graph with hubs (8 192 vertices, 32 stored entries per row on average), width 64
    "rmat_k4_f64": {"synth": {"workload": 8192, "nnz": 262144, "seed": 3, "k": 4, "L": 3, "f": 64}, "P": 4, "seed": 2025},
    "gemat11p_k3_f32": {"write": {"mtx": "gemat11p.A.mtx", "partvec": "gemat11.mtx.3.hp", "k": 3, "L": 3, "f": 32,
                                  "value_format": "%.9g"}, "P": 3, "seed": 99},
}


def materialise(case: dict, tmp: str) -> str:
    """The data directory of a case (shared with tests/test_reference_grbgcn.py)."""
    if "dir" in case and "config" in case:                # the committed directory under another config file
        out = os.path.join(tmp, "data")
        shutil.copytree(os.path.join(HERE, case["dir"]), out)
        with open(os.path.join(out, "config"), "w") as fh:
            fh.write("%d %s \n" % (len(case["config"]) - 1, " ".join(str(x) for x in case["config"])))
        return out
    if "dir" in case:
        return os.path.join(HERE, case["dir"])
    if "tar" in case:
        with tarfile.open(os.path.join(HERE, case["tar"])) as tf:
            tf.extractall(tmp)
        (sub,) = [d for d in os.listdir(tmp) if os.path.isdir(os.path.join(tmp, d))]
        return os.path.join(tmp, sub)
    if "synth" in case:
        w = case["synth"]
        synth = pkg("synth")
        n, row, col, val = synth.make_graph(w["workload"], w.get("nnz"), seed=w["seed"])
        A = sp.csr_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
        pv = synth.random_partvec(n, w["k"], seed=w["seed"]).numpy() if w["k"] > 1 else np.zeros(n, np.int64)
        out = os.path.join(tmp, "data")
        pkg("pargcn_io").write_directory(out, A, pv, w["k"], w["L"], w["f"], value_format="%.9g")
        return out
    w = case["write"]
    A = sp.csr_matrix(mmread(os.path.join(HERE, w["mtx"]))).astype(np.float32)
    out = os.path.join(tmp, "data")
    pkg("pargcn_io").write_directory(out, A, read_partvec(os.path.join(HERE, w["partvec"])), w["k"], w["L"], w["f"],
                                     value_format=w["value_format"])
    return out


def read_dump(path: str):
    """Matrices alive at GrB_finalize, in creation order (format: oracle/shim/grb_shim.c)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    assert buf[:4] == b"GRBD"
    (count,), at, out = struct.unpack_from("<q", buf, 4), 12, []
    for _ in range(count):
        serial, nrows, ncols, nvals = struct.unpack_from("<4q", buf, at); at += 32
        I = np.frombuffer(buf, np.int64, nvals, at); at += 8 * nvals
        J = np.frombuffer(buf, np.int64, nvals, at); at += 8 * nvals
        X = np.frombuffer(buf, np.float32, nvals, at); at += 4 * nvals
        out.append((serial, nrows, ncols, I, J, X))
    assert at == len(buf)
    return out


def run_reference(directory: str, P: int, seed: int, tmp: str):
    env = dict(os.environ, MPISHIM_NP=str(P), MPISHIM_SEED=str(seed), GRBSHIM_DUMP=os.path.join(tmp, "dump"))
    res = subprocess.run([BIN, "-p", directory, "-c", os.path.join(directory, "config"), "-t", "1"], env=env, text=True,
                         capture_output=True, timeout=600)
    assert res.returncode == 0 and "Graphblas error" not in res.stderr, (res.returncode, res.stderr[-2000:])
    lines = res.stdout.strip().split("\n")
    errs = [x for x in re.findall(r"^err:(\S+)$", res.stdout, re.M)]
    stats = [int(x) for x in lines[-1].split()]
    assert len(errs) == 3 and len(stats) == 8, res.stdout
    L, d = pkg("pargcn_io").read_config(os.path.join(directory, "config"))
    finals = []
    for r in range(P):
        W = {}
        # the weight matrices are the only survivors with fewer than n rows; W[1] .. W[L-1] were created in that order
        mats = [m for m in read_dump(os.path.join(tmp, "dump.%d" % r)) if m[1] != d[0]]
        assert len(mats) == L - 1
        for l, (_, nrows, ncols, I, J, X) in enumerate(mats, start=1):
            assert (nrows, ncols) == (d[l], d[l + 1]) and X.size == nrows * ncols          # dense
            W[l] = np.zeros((nrows, ncols), np.float32)
            W[l][I, J] = X
        finals.append(W)
    for r in range(1, P):
        for l in finals[0]:
            assert np.array_equal(finals[0][l], finals[r][l]), "ranks disagree on W[%d]" % l
    return res.stdout, errs, stats, finals[0], L, d


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    pargcn = pkg("pargcn")
    for name, case in CASES.items():
        with tempfile.TemporaryDirectory() as tmp:
            directory = materialise(case, tmp)
            stdout, errs, stats, Wend, L, d = run_reference(directory, case["P"], case["seed"], tmp)
            again = run_reference(directory, case["P"], case["seed"], tmp)                    # reproducible?
            assert again[1] == errs and all(np.array_equal(again[3][l], Wend[l]) for l in Wend)
        W0 = pargcn.init_weights(d, case["seed"], "glibc")
        arrays = {}
        for l in range(1, L):
            arrays["w0_%d" % l] = W0[l]
            arrays["w_%d" % l] = Wend[l]
        np.savez_compressed(os.path.join(HERE, "pargcn_ref_%s.npz" % name), **arrays)
        meta = dict(case, L=L, d=d, err_printed=errs, stats=stats, stdout=stdout.split("\n"),
                    source="/root/reference/Parallel-GCN/main.c, unmodified, built by `make -C oracle ref` "
                           "(GraphBLAS / MPI stand-ins: oracle/shim/)")
        meta["stdout"] = [ln for ln in meta["stdout"] if not ln.startswith("time :")]
        with open(os.path.join(HERE, "pargcn_ref_%s.json" % name), "w") as fh:
            json.dump(meta, fh, indent=1)
        print(name, "P=%d" % case["P"], "d=%s" % d, "err", errs, "stats", stats)


if __name__ == "__main__":
    main()
