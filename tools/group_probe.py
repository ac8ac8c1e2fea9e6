#!/usr/bin/env python3
"""Run ONE launch group of the aggregation -- the forward SpMM of one block of one rank -- a few times and nothing
else, built exactly as bench.py builds it (bench.acquire_partition): what the rocprofv3 --pmc passes attribute
their counters to (tools/final_profile.sh).

    python tools/group_probe.py [--workload reddit] [--generator rmat] [--partvec random] [--emulate-rank r/P]
                                [--features 128] [--block loc|halo0|halo1] [--reps 3]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--generator", default="rmat")
    ap.add_argument("--partvec", default="random")
    ap.add_argument("--emulate-rank", default=None)
    ap.add_argument("--features", type=int, default=None)
    ap.add_argument("--block", default="loc")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--shards", default=None)
    ap.add_argument("--mtx", default=None)
    ap.add_argument("--real", action="store_true")
    a = ap.parse_args()
    synth, engine, kernels = bench.pkg("synth"), bench.pkg("engine"), bench.pkg("kernels")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    f = a.features or synth.SHAPES[a.workload][2]
    part, info = bench.acquire_partition(a, 0, 1, dev, lambda m: None, with_transpose=False)
    K = kernels.HipKernels(dev)
    eng = engine.AggregationEngine(part, K, dev, bench.NoExchange() if part.size > 1 else None)
    A = eng.A_loc if a.block == "loc" else eng.A_halo[int(a.block[4:])]
    B = torch.rand(A.ncols, f, device=dev) * 2 - 1
    C = torch.zeros(part.n_local, f, device=dev)
    for _ in range(a.reps):
        K.spmm(A, B, C, accumulate=a.block != "loc")
    torch.cuda.synchronize()
    print("block %s: nnz %d gather %d strip %d dense %d core %d alg_bytes %d" % (
        a.block, A.nnz, A.col.numel(), A.strip.nnz if A.strip else 0, A.dense3.nnz if A.dense3 else 0, A.core.nnz if A.core else 0,
        A.alg_bytes(f)))


if __name__ == "__main__":
    main()
