#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the committed outputs
are what travels to the GPU box.  Usage:  python tests/golden/make_golden.py

What is produced (all by executing the reference's own code, unmodified,
imported from /root/reference/GPU/PGCN.py under the gloo backend on CPU):

  fixtures   karate.mtx / gemat11.mtx + the shipped part vectors, copied as DATA
             (Newman/karate and HB/gemat11 from the SuiteSparse collection);
             the two pickled karate part vectors are re-written in the one-line
             text format PGCN.py -p expects (PGCN.py:172-173).
             *.A.mtx = output of the reference's preprocess/GrB-GNN-IDG.py.
  maps       compute_communication_maps (PGCN.py:37-51) per rank.
  spmm       PSpMM.forward / .backward (PGCN.py:121-134) on seeded inputs with
             non-owned rows zeroed (sidesteps reference quirk Q1);
             backward only for P<=2 (quirk Q3: assignment instead of add).
  train      the body of run() (PGCN.py:186-226) at P=1 with seeded weights:
             per-epoch loss and final weights; cross-checked against the stdout
             of the unmodified ref.run() under the same seed.
"""
import contextlib
import importlib.util
import io
import json
import os
import pickle
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F
from scipy.io import mmread

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
F_SPMM = 4
SEED = 20260921


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_PGCN", os.path.join(REF, "GPU", "PGCN.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def read_partvec(path):
    with open(path) as f:
        return list(map(int, f.readline().split()))


def _setup_rank(ref, rank, P, path_A, path_pv, f):
    """PGCN.py:163-189, the setup part of run(), line by line."""
    ref.myrank, ref.world_size = rank, P
    ref.device = torch.device("cpu")
    A = mmread(path_A)
    partvec = read_partvec(path_pv)
    ref.send_map, ref.recv_map = ref.compute_communication_maps(A, partvec, rank, P)
    Ap = ref.get_partitiont_of_adjacency_matrix(A, partvec, rank)
    ref.send_buffers, ref.recv_buffers = {}, {}
    for (source, indices) in ref.recv_map.items():
        ref.recv_buffers[source] = torch.zeros(len(indices), f)
    for (target, indices) in ref.send_map.items():
        ref.send_buffers[target] = torch.zeros(len(indices), f)
    ref.init_stats()
    ref.X = torch.zeros(A.shape[0], f)
    return A, partvec, Ap


def _worker(rank, P, port, path_A, path_pv, f, do_bwd, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=P)
    ref = load_ref()
    A, partvec, Ap = _setup_rank(ref, rank, P, path_A, path_pv, f)
    n = A.shape[0]
    own = np.asarray(partvec) == rank
    rng = np.random.default_rng(SEED)
    Hfull = (rng.random((n, f), dtype=np.float32) * 2 - 1)
    Gfull = (rng.random((n, f), dtype=np.float32) * 2 - 1)
    Hin = torch.tensor(Hfull * own[:, None], requires_grad=True)
    out = ref.PSpMM.apply(Ap, Hin)
    res = {
        "rank": rank,
        "n_local": int(own.sum()),
        "nnz_local": int(Ap._nnz()),
        "send_map": {int(k): v.numpy().copy() for k, v in ref.send_map.items()},
        "recv_map": {int(k): v.numpy().copy() for k, v in ref.recv_map.items()},
        "fwd_owned": out.detach().numpy()[own].copy(),
        "stats_fwd": {k: int(v) for k, v in ref.stats.items()},
    }
    if do_bwd:
        ref.X.zero_()  # sidestep quirk Q2 (stale X)
        out.backward(torch.tensor(Gfull * own[:, None]))
        res["bwd_owned"] = Hin.grad.numpy()[own].copy()
    out_q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def run_case(name, path_A, path_pv, P, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    do_bwd = P <= 2
    procs = [ctx.Process(target=_worker, args=(r, P, port, path_A, path_pv, F_SPMM, do_bwd, q))
             for r in range(P)]
    for p in procs:
        p.start()
    results = [q.get() for _ in range(P)]
    for p in procs:
        p.join()
    results.sort(key=lambda r: r["rank"])
    partvec = np.asarray(read_partvec(path_pv))
    n = partvec.shape[0]
    fwd = np.zeros((n, F_SPMM), np.float32)
    bwd = np.zeros((n, F_SPMM), np.float32) if do_bwd else None
    arrays, meta = {}, {"P": P, "f": F_SPMM, "seed": SEED, "ranks": []}
    for r in results:
        own = partvec == r["rank"]
        fwd[own] = r["fwd_owned"]
        if do_bwd:
            bwd[own] = r["bwd_owned"]
        meta["ranks"].append({
            "rank": r["rank"], "n_local": r["n_local"], "nnz_local": r["nnz_local"],
            "send": {str(k): int(v.size) for k, v in r["send_map"].items()},
            "recv": {str(k): int(v.size) for k, v in r["recv_map"].items()},
            "stats_fwd": r["stats_fwd"]})
        for k, v in r["send_map"].items():
            arrays["send_%d_%d" % (r["rank"], k)] = v.astype(np.int32)
        for k, v in r["recv_map"].items():
            arrays["recv_%d_%d" % (r["rank"], k)] = v.astype(np.int32)
    arrays["fwd"] = fwd
    if do_bwd:
        arrays["bwd"] = bwd
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as fjs:
        json.dump(meta, fjs, indent=1, sort_keys=True)
    print("wrote", name, meta["ranks"])


def train_case(name, path_A, path_pv, nlayers, f, port):
    """P=1: the body of run() PGCN.py:186-226 with the reference's own objects."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=0, world_size=1)
    ref = load_ref()
    A, partvec, Ap = _setup_rank(ref, 0, 1, path_A, path_pv, f)
    n = A.shape[0]
    x = np.vstack([[e] * f for e in range(n)])  # :186-187
    H = torch.tensor(x, dtype=torch.float32, requires_grad=True)  # :188
    labels = torch.arange(0, n) % f  # :192
    torch.manual_seed(SEED)
    model = nn.Sequential(*[ref.PGCN(Ap, f, f) for _ in range(nlayers)])  # :194-196
    ref.initiliaze_parameters(model)  # :199
    w0 = [m.linear.weight.detach().numpy().copy() for m in model]
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)  # :200
    losses = []
    for epoch in range(5):  # 1 warm-up (:202-209) + 4 timed (:212-220)
        logits = model(H)
        logp = F.log_softmax(logits, 1)
        loss = F.nll_loss(logp, labels)
        optimizer.zero_grad()
        loss.backward()
        ref.average_gradients(model)
        optimizer.step()
        losses.append(float(loss))
    w1 = [m.linear.weight.detach().numpy().copy() for m in model]
    total_vol, total_nmsg = int(ref.stats["send_volume"]), int(ref.stats["send_nmsg"])

    # cross-check the driver above against the UNMODIFIED ref.run() under the same seed
    torch.manual_seed(SEED)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref.run(0, 1, nlayers, f, path_A, path_pv, "gloo")
    printed = [float(l.split("Loss")[1]) for l in buf.getvalue().splitlines() if "Loss" in l]
    for a, b in zip(printed, losses[1:]):
        assert abs(a - b) <= 5.1e-5 * max(1.0, abs(b)) + 5.1e-5, (printed, losses)
    dist.destroy_process_group()

    arrays = {"losses": np.asarray(losses, np.float64)}
    for i, (a, b) in enumerate(zip(w0, w1)):
        arrays["w0_%d" % i] = a
        arrays["w1_%d" % i] = b
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    with open(os.path.join(OUT, name + ".json"), "w") as fjs:
        json.dump({"nlayers": nlayers, "f": f, "seed": SEED, "losses": losses,
                   "ref_run_stdout_losses": printed, "total_vol": total_vol,
                   "total_nmsg": total_nmsg}, fjs, indent=1, sort_keys=True)
    print("wrote", name, losses, printed)


def main():
    # ---- fixtures (data, not source) ----
    shutil.copy(os.path.join(REF, "GPU/SHP/data/karate/karate.mtx"), os.path.join(OUT, "karate.mtx"))
    shutil.copy(os.path.join(REF, "GPU/hypergraph/data/gemat11/gemat11.mtx"),
                os.path.join(OUT, "gemat11.mtx"))
    for ext in ("hp", "rp"):
        shutil.copy(os.path.join(REF, "GPU/hypergraph/data/gemat11.mtx.3." + ext),
                    os.path.join(OUT, "gemat11.mtx.3." + ext))
    for ext in ("hp", "stchp"):
        with open(os.path.join(REF, "GPU/SHP/data/partvec.%s.3" % ext), "rb") as f:
            pv = pickle.load(f)
        with open(os.path.join(OUT, "karate.mtx.3." + ext), "w") as f:
            f.write("".join("%d " % p for p in pv) + "\n")  # GPU/hypergraph/main.cpp:51-63 format
    # 2-way and 1-way part vectors (seeded) for the P=2 backward and P=1 training cases
    rng = np.random.default_rng(SEED)
    for nm, n in (("karate", 34), ("gemat11", 4929)):
        with open(os.path.join(OUT, "%s.mtx.2.rp" % nm), "w") as f:
            f.write("".join("%d " % p for p in rng.integers(0, 2, n)) + "\n")
        with open(os.path.join(OUT, "%s.mtx.1.rp" % nm), "w") as f:
            f.write("0 " * n + "\n")
    # normalised adjacency by the reference's own preprocess script
    # (gemat11 holds negative values, so its row sums are not normalisable: the
    #  reference's preprocess yields NaN on it.  Use its PATTERN, written as gemat11p.mtx.)
    from scipy.io import mmwrite
    import scipy.sparse as sp
    G = sp.coo_matrix(mmread(os.path.join(OUT, "gemat11.mtx")))
    mmwrite(os.path.join(OUT, "gemat11p.mtx"),
            sp.coo_matrix((np.ones(G.nnz, np.int64), (G.row, G.col)), shape=G.shape), field="pattern")
    with tempfile.TemporaryDirectory() as td:
        for nm in ("karate", "gemat11p"):
            shutil.copy(os.path.join(OUT, nm + ".mtx"), os.path.join(td, nm + ".mtx"))
            subprocess.check_call([sys.executable, os.path.join(REF, "preprocess/GrB-GNN-IDG.py"),
                                   "-i", os.path.join(td, nm + ".mtx"), "-f", "16", "-l", "3"],
                                  stdout=subprocess.DEVNULL)
            shutil.copy(os.path.join(td, nm + ".A.mtx"), os.path.join(OUT, nm + ".A.mtx"))

    port = 29710
    cases = [
        ("ref_karate_hp3", "karate.mtx", "karate.mtx.3.hp", 3),
        ("ref_karate_stchp3", "karate.mtx", "karate.mtx.3.stchp", 3),
        ("ref_karate_rp2", "karate.mtx", "karate.mtx.2.rp", 2),
        ("ref_karate_p1", "karate.mtx", "karate.mtx.1.rp", 1),
        ("ref_gemat11_hp3", "gemat11.mtx", "gemat11.mtx.3.hp", 3),
        ("ref_gemat11_rp3", "gemat11.mtx", "gemat11.mtx.3.rp", 3),
        ("ref_gemat11_rp2", "gemat11.mtx", "gemat11.mtx.2.rp", 2),
        ("ref_gemat11pA_rp2", "gemat11p.A.mtx", "gemat11.mtx.2.rp", 2),
        ("ref_gemat11pA_hp3", "gemat11p.A.mtx", "gemat11.mtx.3.hp", 3),
    ]
    for name, a, pv, P in cases:
        port += 1
        run_case(name, os.path.join(OUT, a), os.path.join(OUT, pv), P, port)
    train = [
        ("ref_train_karate", "karate.mtx", "karate.mtx.1.rp", 2, 16),
        ("ref_train_karateA", "karate.A.mtx", "karate.mtx.1.rp", 3, 16),
        ("ref_train_gemat11pA", "gemat11p.A.mtx", "gemat11.mtx.1.rp", 3, 16),
    ]
    ctx = mp.get_context("spawn")
    for name, a, pv, L, f in train:
        port += 1
        p = ctx.Process(target=train_case,
                        args=(name, os.path.join(OUT, a), os.path.join(OUT, pv), L, f, port))
        p.start()
        p.join()
        assert p.exitcode == 0, name


if __name__ == "__main__":
    main()
