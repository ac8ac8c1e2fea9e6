#!/usr/bin/env python3
"""r06 probe: does the ROW STRIDE of the feature panel cost the gather part L2 capacity?  A 64-feature pass over an n x 128 fp32 panel
touches 256 of every 512 bytes: if the L2 indexes its sets with plain address bits, half of the sets never see a row.  The SAME
64-wide product of the benchmark block on (a) the left half of an n x 128 panel (row stride 512 B) and (b) a compact n x 64 panel
(row stride 256 B), the kernels of the launch group one after the other (HIP events per kernel), plus the 32-wide variants.

    python tools/layout_probe.py [--generator rmat] [--reps 10]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--generator", default="rmat")
    ap.add_argument("--partvec", default="random")
    ap.add_argument("--emulate-rank", default=None)
    ap.add_argument("--shards", default=None)
    ap.add_argument("--mtx", default=None)
    ap.add_argument("--real", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    engine, kernels = bench.pkg("engine"), bench.pkg("kernels")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    part, info = bench.acquire_partition(a, 0, 1, dev, lambda m: None, with_transpose=False)
    K = kernels.HipKernels(dev)
    eng = engine.AggregationEngine(part, K, dev, None)
    A = eng.A_loc
    n = A.ncols
    wide = torch.rand(n, 128, device=dev) * 2 - 1
    real_lib = K.lib
    K.single_lane = True
    for w in (64, 32):
        cases = {"strided (row stride 512 B)": wide[:, :w], "compact (row stride %d B)" % (4 * w): wide[:, :w].contiguous()}
        for name, B in cases.items():
            C = torch.zeros(part.n_local, w, device=dev)
            st = bench.SplitTimer(real_lib)
            K.lib = st
            A.launch_cache.clear()
            for _ in range(a.reps + 2):
                K.spmm(A, B, C)
            torch.cuda.synchronize()
            K.lib = real_lib
            A.launch_cache.clear()
            split = {k: round(v, 1) for k, v in st.summary_us().items()}
            split["sum"] = round(sum(split.values()), 1)
            print("f = %3d  %-28s %s" % (w, name, split), flush=True)


if __name__ == "__main__":
    main()
