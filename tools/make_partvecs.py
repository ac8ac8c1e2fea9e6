#!/usr/bin/env python3
"""Part vectors from the REFERENCE's own partitioner front-ends for a synthetic graph.

Builds /root/reference/GPU/hypergraph/main.cpp (PaToH column-net model, writes <name>.<k>.hp and .rp) and
/root/reference/GPU/graph/main.cpp (METIS, writes .gp) with the commands of SURVEY App. C into a scratch
directory (the reference tree is read-only, nothing is copied into the repo), writes the seeded synthetic
graph of `bench.py --workload mid` as a MatrixMarket file, runs both tools for every k and keeps ONLY the part
vectors (tests/golden/partvec/) plus a statistics file: exchange volume (boundary rows per aggregation) of the
random / hypergraph / graph partitions as counted by this engine's partition code.

Needs /root/reference: runs in the build container only; the outputs are committed.
usage: python tools/make_partvecs.py [--workload mid] [--k 2,4,8]"""
import argparse
import importlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import scipy.sparse as sp
import torch
from scipy.io import mmwrite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
REF = "/root/reference"


def build_tools(work):
    hp, gp = os.path.join(work, "gcnhgp"), os.path.join(work, "gcngp")
    if not os.path.exists(hp):
        subprocess.check_call(["g++", "-fopenmp", "-O3", "-std=c++11", "-w", "-I", REF + "/GCN-HP/lib/Linux-x86_64",
                               REF + "/GPU/hypergraph/main.cpp", "-o", hp, "-L", REF + "/GCN-HP/lib/Linux-x86_64", "-lpatoh"])
    if not os.path.exists(gp):
        subprocess.check_call(["g++", "-fopenmp", "-O3", "-std=c++11", "-w", "-I", REF + "/GCN-GP/lib/include",
                               REF + "/GPU/graph/main.cpp", "-o", gp, "-L", REF + "/GCN-GP/lib/lib", "-lmetis"])
    return hp, gp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="mid")
    ap.add_argument("--k", default="2,4,8")
    ap.add_argument("--work", default="/tmp/pgcn_partvec")
    ap.add_argument("--generator", default="rmat", choices=["rmat", "sbm", "shardstream"],
                    help="sbm: the planted-partition stand-in (portable stream: the same graph on the GPU box); "
                         "shardstream: the union of the rank-local shards tools/make_shards.py generates "
                         "(synth.rmat_shard_keys: papers-scale workloads, with --scale)")
    ap.add_argument("--scale", type=float, default=1.0, help="shardstream: shrink n and nnz of the workload by this factor")
    ap.add_argument("--tools", default="hp,gp", help="which of the reference's front-ends to run")
    ap.add_argument("--graph-name", default=None, help="a synth.SHAPES name, or (with --n/--nnz) a free label")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--nnz", type=int, default=None, help="directed entries without self loops")
    args = ap.parse_args()
    synth = importlib.import_module(PKG + ".synth")
    partition = importlib.import_module(PKG + ".partition")
    io_ = importlib.import_module(PKG + ".pargcn_io")
    os.makedirs(args.work, exist_ok=True)
    hp, gp = build_tools(args.work)
    tools = args.tools.split(",")
    if args.generator == "shardstream":
        n0, nnz0, _, _ = synth.SHAPES[args.workload]
        n, pairs = max(64, int(n0 * args.scale)), max(64, int(nnz0 * args.scale) // 2)       # as tools/make_shards.py
        keys = synth.rmat_shard_keys(n, pairs, 0, torch.zeros(n, dtype=torch.int64), seed=0)
        row, col, val = synth.shard_normalize(n, keys, torch.bincount(keys // n, minlength=n))
    elif args.n:
        n, row, col, val = synth.make_graph(args.n, args.nnz, seed=0, generator=args.generator)
    else:
        n, row, col, val = synth.make_graph(args.workload, seed=0, generator=args.generator)
    label = args.graph_name or (args.workload + ("-sbm" if args.generator == "sbm" else ""))
    name = "%s.A.mtx" % label
    print("graph %s: n=%d entries=%d" % (label, n, row.numel()), flush=True)
    mtx = os.path.join(args.work, name)
    A = sp.coo_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    mmwrite(mtx, sp.tril(A).tocoo(), symmetry="symmetric", precision=3)        # like preprocess/GrB-GNN-IDG.py:80
    del A
    print("wrote", mtx, flush=True)
    outdir = os.path.join(ROOT, "tests", "golden", "partvec")
    os.makedirs(outdir, exist_ok=True)
    stats = {"workload": label, "generator": args.generator, "n": n, "nnz": int(row.numel()), "parts": {}}
    for k in [int(x) for x in args.k.split(",")]:
        o = os.path.join(args.work, "out%d" % k) + "/"
        shutil.rmtree(o, ignore_errors=True)
        os.makedirs(o)
        if "hp" in tools:
            subprocess.check_call([hp, "-a", mtx, "-o", o, "-k", str(k)], stdout=subprocess.DEVNULL)
            print("hp done", k, flush=True)
        if "gp" in tools:
            subprocess.check_call([gp, "-a", mtx, "-o", o, "-k", str(k)], stdout=subprocess.DEVNULL)
            print("gp done", k, flush=True)
        rec = {}
        for ext in [t for t in ("hp", "gp") if t in tools] + ["rp"]:
            src = os.path.join(o, "%s.%d.%s" % (name, k, ext))
            dst = os.path.join(outdir, "%s.%d.%s" % (name, k, ext))
            if ext == "rp":          # the reference seeds its random vector with the clock: keep OUR seeded one instead
                io_.write_partvec(dst, synth.random_partvec(n, k, seed=0).numpy())
            else:
                shutil.copyfile(src, dst)
            pv = torch.tensor(partition.read_partvec(dst), dtype=torch.int64)
            assert pv.numel() == n and int(pv.max()) < k
            rows = 0
            sizes = torch.bincount(pv, minlength=k)
            nnz_p = torch.bincount(pv[row], minlength=k)
            prow, pcol = pv[row], pv[col]
            cut = prow != pcol
            # boundary rows = unique (needing part, column) pairs: rows of H that travel per aggregation
            rows = int(torch.unique(prow[cut] * n + col[cut]).numel())
            rec[ext] = {"boundary_rows_per_aggregation": rows, "max_part_vertices": int(sizes.max()),
                        "max_part_nnz": int(nnz_p.max()), "imbalance_nnz": float(nnz_p.max()) * k / float(row.numel())}
        stats["parts"][str(k)] = rec
        print(k, json.dumps(rec))
    with open(os.path.join(outdir, "%s.stats.json" % label), "w") as fh:
        json.dump(stats, fh, indent=1)


if __name__ == "__main__":
    main()
