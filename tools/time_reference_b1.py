#!/usr/bin/env python3
"""CPU baseline B1: the UNMODIFIED reference engine /root/reference/GPU/PGCN.py
run with `-b gloo` on the host cores of THIS container, on the `mid` workload (n = 131 072, 4.3 M entries, f = 64,
L = 2 -- the size at which its O(nnz) Python set-up stays in seconds), timed by its own `Elapsed time` line (4 epochs
after one warm-up epoch, PGCN.py:202-228).  Needs /root/reference: build container only; the record is committed
under profiles/ next to the GPU line of the same workload (`python bench.py --workload mid`).

usage: python tools/time_reference_b1.py [--ranks 1] [--threads N] [--out profiles/r03_cpu_baseline_B1_mid.json]"""
import argparse, importlib, json, os, re, socket, subprocess, sys, time
import scipy.sparse as sp
from scipy.io import mmwrite
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
REF = "/root/reference/GPU/PGCN.py"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mid")
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--work", default="/tmp/pgcn_b1")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_cpu_baseline_B1_mid.json"))
    a = ap.parse_args()
    synth, io_ = importlib.import_module(PKG + ".synth"), importlib.import_module(PKG + ".pargcn_io")
    os.makedirs(a.work, exist_ok=True)
    n, nnz_dir, f, L = synth.SHAPES[a.workload]
    mtx = os.path.join(a.work, a.workload + ".A.mtx")
    if not os.path.exists(mtx):
        n, row, col, val = synth.make_graph(a.workload, seed=0)
        mmwrite(mtx, sp.coo_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n)), precision=9)
    with open(mtx) as fh:
        while True:
            line = fh.readline()
            if not line.startswith("%"):
                break
    stored = int(line.split()[2])
    symmetric = "symmetric" in open(mtx).readline()
    pv = os.path.join(a.work, "%s.%d.rp" % (a.workload, a.ranks))
    io_.write_partvec(pv, synth.random_partvec(n, a.ranks, seed=0).numpy() if a.ranks > 1 else [0] * n)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    procs, t0 = [], time.time()
    for r in range(a.ranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(a.ranks), SLURM_NPROCS=str(a.ranks),
                   SLURM_PROCID=str(r), OMP_NUM_THREADS=str(max(1, a.threads // a.ranks)))
        procs.append(subprocess.Popen([sys.executable, REF, "-a", mtx, "-p", pv, "-b", "gloo", "-s", str(a.ranks), "-l", str(L),
                                       "-f", str(f)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate()[0] for p in procs]
    wall = time.time() - t0
    assert all(p.returncode == 0 for p in procs), outs[0][-2000:]
    elapsed = float(re.search(r"Elapsed time ([0-9.]+)", outs[0]).group(1))
    nnz = stored * 2 - n if symmetric else stored                      # stored entries of the full matrix (self loops once)
    rec = {"baseline": "B1: unmodified /root/reference/GPU/PGCN.py -b gloo (torch.sparse.mm on CPU)", "workload": a.workload,
           "n": n, "nnz": nnz, "f": f, "layers": L, "ranks": a.ranks, "omp_threads_per_rank": max(1, a.threads // a.ranks),
           "host_cpus": os.cpu_count(), "elapsed_4_epochs_s": elapsed, "ms_per_epoch": 1e3 * elapsed / 4,
           "edges_per_s": 2 * L * nnz / (elapsed / 4), "wall_s_incl_setup": wall,
           "losses": re.findall(r"Epoch \d+ \| Loss ([0-9.]+)", outs[0]),
           "note": "timed in the build container (the GPU box has no /root/reference); the reference's own Elapsed-time line; "
                   "its set-up (mmread + the Python loop of compute_communication_maps, PGCN.py:37-51) is in wall_s_incl_setup"}
    with open(a.out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
