#!/usr/bin/env python3
"""Trace-driven L2 model of the gather part (tools/l2sim/l2sim.c): hit rate and fabric reads of feature rows for
task orders / slice counts / feature passes, WITHOUT a GPU.  Input: the gather part of a plan as saved by
`--dump` (rowptr, col, slice_cnt, row_flags of partition.csr_from_coo on the degree-sorted benchmark graph).

  python tools/l2sim/run_l2sim.py --dump /tmp/gather_part.npz          # build the benchmark graph on the CPU (~10 min)
  python tools/l2sim/run_l2sim.py /tmp/gather_part.npz                 # the model's table

Calibration (r02, MI355X PMC, profiles/r02_pmc_final.txt and gpurun r02_p33): whole rows: L2 hit 51 %, 5.40 GB of
fabric reads by the gather kernel; 64-feature passes: 62 %, 4.46 GB; 32-feature passes: 57 %, 6.05 GB (incl. the
re-read (col, val) pairs, which the model leaves out: 0.17 GB per pass)."""
import argparse
import ctypes
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
pkg = lambda s: importlib.import_module(PKG + "." + s)


def dump(path):
    import torch
    synth, partition = pkg("synth"), pkg("partition")
    n, row, col, val = synth.make_graph("reddit", seed=0, device=torch.device("cpu"))
    deg = torch.bincount(row, minlength=n) + torch.bincount(col, minlength=n)
    rank = torch.empty(n, dtype=torch.int64)
    rank[torch.argsort(-deg, stable=True)] = torch.arange(n)
    h = partition.csr_from_coo(rank[row], rank[col], val, n, n, nslices=8, core=True)
    np.savez(path, rowptr=h.rowptr.numpy(), col=h.col.numpy(), slice_cnt=h.slice_cnt.numpy(), row_flags=h.row_flags.numpy())


def load_sim():
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "l2sim.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "l2sim.c")):
        os.system("gcc -O2 -shared -fPIC -o %s %s" % (so, os.path.join(here, "l2sim.c")))
    L = ctypes.CDLL(so)
    L.l2sim_slice.restype = ctypes.c_int
    L.l2sim_slice.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 2
    return L


def simulate(L, tasks_by_slice, col, window, passes, lines_per_pass, l2_bytes=4 << 20, ways=16, batch=8):
    sets = l2_bytes // 128 // ways
    hits = misses = 0
    for t in tasks_by_slice:
        t = np.ascontiguousarray(t, dtype=np.int64)
        h, m = ctypes.c_int64(), ctypes.c_int64()
        rc = L.l2sim_slice(t.ctypes.data, t.shape[0], col.ctypes.data, window, batch, passes, lines_per_pass, sets, ways,
                           ctypes.byref(h), ctypes.byref(m))
        assert rc == 0
        hits += h.value
        misses += m.value
    return hits, misses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("part", nargs="?", default="/tmp/gather_part.npz")
    ap.add_argument("--dump", default=None)
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--small-row", type=int, default=96)
    args = ap.parse_args()
    if args.dump:
        dump(args.dump)
        return
    kernels = pkg("kernels")
    z = np.load(args.part)
    rowptr, col = z["rowptr"], np.ascontiguousarray(z["col"], dtype=np.int32)
    tasks, fix, nslots, seg = kernels.build_plan(rowptr, args.chunk, z["slice_cnt"], args.small_row, row_flags=z["row_flags"])
    seg = [int(seg[i]) for i in range(9)]
    kbeg = (tasks[:, 0].astype(np.int64) & 0xffffffff) | (tasks[:, 1].astype(np.int64) << 32)
    t2 = np.stack([kbeg, tasks[:, 2].astype(np.int64)], 1)
    by_slice = [t2[seg[s]:seg[s + 1]] for s in range(8)]
    nnz = int(col.shape[0])
    print("gather part: %d entries, %d tasks (%.1f entries per task), %d partial-sum slots" % (nnz, t2.shape[0], nnz / t2.shape[0], nslots))
    L = load_sim()
    rng = np.random.default_rng(0)

    def report(name, tb, window, passes, lpp, **kw):
        t0 = time.time()
        h, m = simulate(L, tb, col, window, passes, lpp, **kw)
        print("%-58s hit %5.1f %%  fabric reads of feature rows %.2f GB   (%.0f s)" % (name, 100.0 * h / (h + m), m * 128 / 1e9, time.time() - t0), flush=True)

    # the shipped orders: longest first per slice; 32 CUs x 6 workgroups x 8 tasks resident per XCD (16 tasks per workgroup in passes)
    report("whole rows, longest first (measured: 51 %, ~5.2 GB)", by_slice, 1536, 1, 4)
    report("64-feature passes (measured: 62 %, ~4.3 GB)", by_slice, 3072, 2, 2)
    report("32-feature passes (measured: 57 % incl. pair re-reads)", by_slice, 6144, 4, 1)
    # what-ifs
    report("whole rows, half the resident tasks", by_slice, 768, 1, 4)
    report("whole rows, 2x / 4x the resident tasks (= passes that run CONCURRENTLY)", by_slice, 3072, 1, 4)
    report("   ... 4x", by_slice, 6144, 1, 4)
    report("whole rows, random task order", [t[rng.permutation(t.shape[0])] for t in by_slice], 1536, 1, 4)
    first_col = [col[t[:, 0]] for t in by_slice]
    report("whole rows, tasks ordered by their first column", [t[np.argsort(fc, kind="stable")] for t, fc in zip(by_slice, first_col)], 1536, 1, 4)
    report("whole rows, an 8 MB L2 (what 16 time slices would buy at best)", by_slice, 1536, 1, 4, l2_bytes=8 << 20)
    report("64-feature passes, an 8 MB L2", by_slice, 3072, 2, 2, l2_bytes=8 << 20)


if __name__ == "__main__":
    main()
