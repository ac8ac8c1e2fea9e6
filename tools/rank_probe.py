#!/usr/bin/env python3
"""Per-rank compute of an N-GPU run, measured on ONE GPU: build rank r's partition of the benchmark
graph for world size P and time its forward / backward aggregation with a no-op exchanger (the halo
slab holds random rows).  Shows what the compute side of bench.py --gpus P costs per rank."""
import argparse, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
pkg = lambda s: importlib.import_module(PKG + "." + s)

class NoExchange:
    name = "none"
    def alltoallv(self, *a): pass
    def allreduce_sum(self, b): pass

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--f", type=int, default=128); ap.add_argument("--rounds", type=int, default=10)
    a = ap.parse_args()
    synth, partition, engine, kernels = pkg("synth"), pkg("partition"), pkg("engine"), pkg("kernels")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    n, row, col, val = synth.make_graph("reddit", seed=0, device=dev)
    pv = synth.random_partvec(n, a.world, seed=0)
    t0 = time.time(); p = partition.build_partition(row, col, val, n, pv, a.rank, a.world); torch.cuda.synchronize(); tb = time.time() - t0
    K = kernels.HipKernels(dev)
    t0 = time.time(); eng = engine.AggregationEngine(p, K, dev, NoExchange() if a.world > 1 else None); torch.cuda.synchronize(); tp = time.time() - t0
    def frac(d):
        ds = d if isinstance(d, list) else [d]
        tot = sum(x.nnz for x in ds if x is not None)
        return 100.0 * sum((x.core.nnz if x.core is not None else 0) + (x.strip.nnz if getattr(x, "strip", None) is not None else 0)
                           + (x.dense3.nnz if getattr(x, "dense3", None) is not None else 0) for x in ds if x is not None) / max(tot, 1)
    print("rounds=%d " % p.rounds, end="")
    print("P=%d rank=%d n_local=%d n_halo=%d n_send=%d nnz_loc=%d nnz_halo=%d | build %.2fs prepare %.2fs | tiled%% (strips + core + MFMA): loc %.0f halo %.0f locT %.0f haloT %.0f"
          % (a.world, a.rank, p.n_local, p.n_halo, p.n_send, p.A_loc.nnz, sum(x.nnz for x in p.A_halo), tb, tp,
             frac(eng.A_loc), frac(eng.A_halo), frac(eng.A_loc_T), frac(eng.A_halo_T)))
    H = torch.rand(p.n_local, a.f, device=dev)
    if a.world > 1:
        eng._slab("halo", eng.n_halo, a.f).uniform_(); eng._slab("send", eng.n_send, a.f).uniform_()
    for name, fn in (("forward", eng.forward), ("backward", eng.backward)):
        fn(H); fn(H); torch.cuda.synchronize(); ts = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(H); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        nnz = p.A_loc.nnz + sum(x.nnz for x in p.A_halo)
        print("  %-8s median %.3f ms  (%.1f G edges/s on this rank; x%d ranks = %.1f G edges/s)" % (name, np.median(ts), nnz / np.median(ts) / 1e6, a.world, a.world * nnz / np.median(ts) / 1e6))

    # full training epoch of this rank (3 layers, Adam), exchange = no-op: compute + host launch cost
    import torch.nn as nn
    P = pkg("PGCN")
    P.device, P.myrank, P.world_size = dev, 0, 1          # world_size 1: no gradient all-reduce
    P.init_stats()
    model = nn.Sequential(*[P.PGCN(eng, a.f, a.f) for _ in range(3)]).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    Hr = torch.rand(p.n_local, a.f, device=dev).requires_grad_(True)
    labels = p.owned.to(dev) % a.f
    def step():
        loss = P.local_loss(model(Hr), labels, n); opt.zero_grad(); loss.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    t_host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 10
    print("  epoch (no exchange): %.2f ms wall, host enqueue %.2f ms" % (1e3 * t_all, 1e3 * t_host))


if __name__ == "__main__":
    main()
