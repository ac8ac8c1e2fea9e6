#!/usr/bin/env python3
"""Full-size parity of ONE rank's aggregation built from a binary CSR shard (BASELINE config 4: the papers100M shape,
rank r of 8, f = 64): tools/make_shards.py --only-rank wrote PREFIX.<r>.pgcsr + PREFIX.degree.npy; this tool builds the
rank's partition exactly as `bench.py --emulate-rank r/P --shards PREFIX` does, runs the forward aggregation
A_loc . H + sum_r A_halo[r] . halo through the HIP engine (features a deterministic function of the GLOBAL vertex id,
halo slab pre-filled: the exchange is not what is tested) and holds a sample of rows -- the heaviest ones and a random
draw -- to a float64 sum over the shard's own CSR entries with the per-row bound of tests/test_fullsize_gpu.py:
    |got_i - ref_i| <= 1e-5 * sum_j |a_ij| |x_j|   element-wise.
Prints one JSON line.   usage: python tools/shard_rank_check.py --shards /tmp/papers --rank 0 --ranks 8 --features 64"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"


def features(ids: torch.Tensor, f: int, dtype) -> torch.Tensor:
    """x[v, j] in [-1, 1): an integer hash of (global id, column) -- the same on any device, for any subset of ids."""
    j = torch.arange(f, dtype=torch.int64, device=ids.device)
    h = (ids[:, None] * 2654435761 + j[None, :] * 40503 + 12345) % 65536
    return (h.to(torch.float64) / 32768.0 - 1.0).to(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", required=True)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--features", type=int, default=64)
    ap.add_argument("--sample", type=int, default=4096)
    a = ap.parse_args()
    ingest, synth, partition, engine, kernels = (importlib.import_module(PKG + "." + m)
                                                 for m in ("ingest", "synth", "partition", "engine", "kernels"))
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    t0 = time.time()
    sh = ingest.read_shard(ingest.shard_path(a.shards, a.rank))
    n = int(sh["n"])
    pv = synth.block_partvec(n, a.ranks)
    r_, c_, v_ = ingest.shard_coo(sh)
    row, col, val = torch.from_numpy(r_).to(dev), torch.from_numpy(c_).to(dev), torch.from_numpy(v_).to(dev)
    deg = torch.from_numpy(np.load(a.shards + ".degree.npy").astype(np.int64)).to(dev)
    meta = json.load(open(a.shards + ".meta.json"))
    part = partition.build_partition_local(row, col, val, n, pv, a.rank, a.ranks,
                                           emulate={"gdeg": 2 * deg, "nnz_global": meta["nnz_global"]})
    del deg
    t_part = time.time() - t0
    K = kernels.HipKernels(dev)
    eng = engine.AggregationEngine(part, K, dev, bench.NoExchange())
    f = a.features
    H = features(part.owned.to(dev), f, torch.float32)
    eng._slab("halo", eng.n_halo, f)[:eng.n_halo] = features(part.halo_global.to(dev), f, torch.float32)
    eng._slab("send", eng.n_send, f)
    C = eng.forward(H)
    torch.cuda.synchronize()
    t_all = time.time() - t0
    # ---- the sample: heaviest local rows (degree order: the first ones) + a seeded random draw ----------------------
    g = torch.Generator().manual_seed(3)
    pick = torch.unique(torch.cat([torch.arange(min(512, part.n_local)), torch.randint(0, part.n_local, (a.sample,), generator=g)]))
    gid = part.owned.cpu()[pick]                                             # global ids of the sampled rows
    rows_sh = torch.from_numpy(sh["rows"])
    pos = torch.searchsorted(rows_sh, gid)                                   # their position in the shard
    assert torch.equal(rows_sh[pos], gid)
    rp = torch.from_numpy(sh["rowptr"])
    beg, cnt = rp[pos], rp[pos + 1] - rp[pos]
    idx = torch.repeat_interleave(beg, cnt) + (torch.arange(int(cnt.sum())) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt))
    seg = torch.repeat_interleave(torch.arange(pick.numel()), cnt).to(dev)
    cols = torch.from_numpy(sh["col"])[idx].to(torch.int64).to(dev)
    vals = torch.from_numpy(sh["val"])[idx].to(dev).to(torch.float64)
    X = features(cols, f, torch.float64)
    ref = torch.zeros((pick.numel(), f), dtype=torch.float64, device=dev).index_add_(0, seg, vals[:, None] * X)
    bound = torch.zeros_like(ref).index_add_(0, seg, vals.abs()[:, None] * X.abs())
    got = C[pick.to(dev)].to(torch.float64)
    worst = float(((got - ref).abs() / (1e-5 * bound + 1e-30)).max())
    rel = float((got - ref).abs().max() / ref.abs().max())
    blocks = {"A_loc": eng.A_loc, **{"A_halo[%d]" % i: b for i, b in enumerate(eng.A_halo)}}
    out = {"check": "one rank's forward aggregation from a binary CSR shard against float64 over the shard's own entries",
           "shards": os.path.basename(a.shards), "rank": a.rank, "ranks": a.ranks, "n": n, "f": f, "n_local": part.n_local,
           "n_halo": part.n_halo, "n_send": part.n_send, "nnz_local": part.nnz_local, "nnz_global": part.nnz_global,
           "rowptr_dtype": str(eng.A_loc.rowptr.dtype), "rows_checked": int(pick.numel()), "entries_checked": int(cnt.sum()),
           "worst_row_error_over_bound_1e-5": worst, "max_rel_err": rel, "passed": bool(worst <= 1.0),
           "blocks": {k: {"nnz": b.nnz, "gather_entries": int(b.col.numel()),
                          "strip": None if b.strip is None else b.strip.nnz, "bf16x3_blocks": None if b.dense3 is None else b.dense3.nnz}
                      for k, b in blocks.items()},
           "partition_s": t_part, "total_s": t_all, "hbm_allocated_GB": torch.cuda.max_memory_allocated(dev) / 2 ** 30}
    print(json.dumps(out))
    sys.exit(0 if worst <= 1.0 else 1)


if __name__ == "__main__":
    main()
