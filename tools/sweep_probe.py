#!/usr/bin/env python3
"""Feasibility probe (r03): would the gather part hit in L2 if every group of lanes OWNED a few rows for the whole
kernel and walked their leftover entries in COLUMN order (all resident groups then sweep the column space together:
the working set at any time is a band of the operand, not all of it)?

The existing one-task-per-row kernel is run on a matrix whose "rows" are m consecutive rows of the gather part
MERGED (entries sorted by column): results are meaningless, timing and L2 counters are exactly those of the access
pattern in question.  m = 1 is the unsliced gather part as it is.  usage: sweep_probe.py [--merge 1,8,32] [--once m]"""
import argparse, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
pkg = lambda s: importlib.import_module(PKG + "." + s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--merge", default="1,4,8,16,32,64", help="0 = balanced packing: consecutive rows until --target entries")
    ap.add_argument("--target", default="1024", help="comma list of entries per merged row for --merge 0")
    ap.add_argument("--once", type=int, default=None)
    ap.add_argument("--f", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--interleave", type=int, default=0, help="merged row = rows r, r + n/m, ... instead of m consecutive ones")
    a = ap.parse_args()
    synth, partition, kernels = pkg("synth"), pkg("partition"), pkg("kernels")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    n, row, col, val = synth.make_graph("reddit", seed=0, device=dev)
    deg = torch.bincount(row, minlength=n) + torch.bincount(col, minlength=n)
    rank = torch.empty(n, dtype=torch.int64, device=dev)
    rank[torch.argsort(-deg, stable=True)] = torch.arange(n, device=dev)
    h = partition.csr_from_coo(rank[row], rank[col], val, n, n, nslices=8, core=True)
    cnt = h.rowptr[1:] - h.rowptr[:-1]
    r = torch.repeat_interleave(torch.arange(n, device=dev), cnt)
    c, v = h.col.to(torch.int64), h.val
    print("gather part: %d entries" % r.numel(), flush=True)
    K = kernels.HipKernels(dev)
    K.fpass = "0"
    K.chunk = 1 << 22
    K.small_row = 1 << 22
    B = torch.rand(n, a.f, device=dev) * 2 - 1
    merges = [a.once] if a.once is not None else [int(x) for x in a.merge.split(",")]
    targets = [int(x) for x in a.target.split(",")]
    cum = torch.cumsum(cnt, 0) - cnt                     # entries ahead of every row
    for m, tgt in [(m, t) for m in merges for t in (targets if m == 0 else [0])]:
        if m == 0:                                       # balanced: merged row = rows whose prefix falls into the same bucket
            rid = cum // tgt
            _, rid = torch.unique_consecutive(rid, return_inverse=True)
            nr = int(rid.max()) + 1
            rm = rid[r]
        else:
            nr = -(-n // m)
            rm = (r % nr) if a.interleave else (r // m)
        hm = partition.csr_from_coo(rm, c, v, nr, n, nslices=1, core=False)
        d = K.prepare(hm)
        C = torch.empty(nr, a.f, device=dev)
        for _ in range(3):
            K.spmm(d, B, C)
        torch.cuda.synchronize()
        if a.once is not None:
            return
        ts = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); K.spmm(d, B, C); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ml = torch.bincount(rm, minlength=nr)
        print("merge %3d target %5d: rows %6d (longest %d)  median %.3f ms  %.1f ps/entry" % (m, tgt, nr, int(ml.max()), np.median(ts), 1e9 * np.median(ts) / r.numel()), flush=True)


if __name__ == "__main__":
    main()
