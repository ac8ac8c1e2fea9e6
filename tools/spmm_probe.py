#!/usr/bin/env python3
"""In-process A/B probe of SpMM variants on the benchmark graph (run under gpurun).
Interleaved rounds, HIP-event timing, median / min per variant (methodology rule 24)."""
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
pkg = lambda s: importlib.import_module(PKG + "." + s)


class TimingLib:
    """Proxy of the C-ABI library: HIP events around every pgcn_spmm_* call (per-kernel split of a launch group)."""

    def __init__(self, lib):
        self._lib, self.rec = lib, []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("pgcn_spmm") or name == "pgcn_spmm_plan_host":
            return fn

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            self.rec.append((name, e0, e1))
            return rc
        return timed

    def summary(self):
        acc = {}
        for name, e0, e1 in self.rec:
            acc.setdefault(name, []).append(e0.elapsed_time(e1))
        return {k: float(np.median(v)) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--f", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--variants", default="s1c1024,s8c1024,s8c512,s8c256,s8c2048")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--libs", default="", help="comma list of variant tags (lib/libpgcn_hip.<tag>.so), A/B in one process")
    ap.add_argument("--split", action="store_true", help="per-kernel split of every variant (HIP events per C-ABI call)")
    ap.add_argument("--once", default=None, help="run one variant once (for rocprofv3 --pmc)")
    args = ap.parse_args()
    synth, partition, kernels = pkg("synth"), pkg("partition"), pkg("kernels")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if args.workload.startswith("uniform:"):
        # ceiling probe: reddit-sized rows (492 entries each) whose columns are uniform in [0, K)
        Kc = int(args.workload.split(":")[1])
        ncols_eff = Kc
        n, deg = 232965, 492
        g = torch.Generator(device=dev); g.manual_seed(0)
        row = torch.arange(n, device=dev).repeat_interleave(deg)
        col = torch.randint(0, Kc, (n * deg,), device=dev, generator=g)
        val = torch.rand(n * deg, device=dev, generator=g)
    else:
        n, row, col, val = synth.make_graph(args.workload, seed=0, device=dev)
        ncols_eff = n
    nnz = row.numel()
    f = args.f
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    B = torch.rand(n, f, device=dev, generator=gen) * 2 - 1
    K = kernels.HipKernels(dev)
    K.fpass = "manual"               # feature passes only where a variant name asks for them (_p64 / _p32)
    variants = {}
    names = [args.once] if args.once else args.variants.split(",")
    _lib = pkg("_lib")
    libs = {"": K.lib}
    if args.libs:
        libs = {t: _lib.load_variant(os.path.join(os.path.dirname(_lib.LIB_PATH), "libpgcn_hip.%s.so" % t))
                for t in args.libs.split(",")}
    prepared = {}
    for name in names:
        # name: s<slices>[g<groups>]c<chunk>[x][k[tau]] (g = column groups; x = xcd swizzle for
        # unsliced; k = degree sort + LDS core)
        name0 = name
        fpass = 0                        # _p64 / _p32: feature passes of the gather part
        if "_p" in name:
            name, fp = name.split("_p")
            fpass = {"64": 16}[fp]
        smode = ""                       # "r:" range slicing (uniform bounds), "d:" dealt order + range slicing
        if ":" in name:
            smode, name = name.split(":")
        head = name[1:name.index("c")]
        G = None
        if "g" in head:
            head, gs = head.split("g")
            G = int(gs)
        S = int(head)
        rest = name[name.index("c") + 1:]
        tau = None
        core = "k" in rest
        if core:
            rest, t = rest.split("k")
            tau = float(t) if t else None
        sw = rest.endswith("x")
        chunk = int(rest.rstrip("x"))
        bounds = None
        if smode in ("r", "d"):
            bounds = torch.tensor([(ncols_eff * i + S - 1) // S for i in range(S + 1)], dtype=torch.int64, device=dev)
        if core:
            deg = torch.bincount(row, minlength=n) + torch.bincount(col, minlength=n)
            rank = torch.empty(n, dtype=torch.int64, device=dev)
            rank[torch.argsort(-deg, stable=True)] = torch.arange(n, device=dev)
            if smode == "d":             # deal the degree order round-robin into S contiguous ranges
                cnt = torch.bincount(torch.arange(n, device=dev) % S, minlength=S)
                bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(cnt, 0)])
                rank = bounds[rank % S] + rank // S
            h = partition.csr_from_coo(rank[row], rank[col], val, n, n, nslices=S, core=True, tau=tau, ngroups=G,
                                       slice_bounds=bounds)
        else:
            h = partition.csr_from_coo(row, col, val, n, ncols_eff, nslices=S, ngroups=G, slice_bounds=bounds)
        K.chunk = chunk
        d = prepared.get(smode + name)
        if d is None:
            d = prepared[smode + name] = K.prepare(h)
        for tag, L in libs.items():
            variants[name0 + ("@" + tag if tag else "")] = (d, sw, L, fpass)
    alg = 8 * nnz + 8 * (n + 1) + 2 * 4 * f * n
    C = torch.empty(n, f, device=dev)

    def run(name):
        d, sw, L, fpass = variants[name]
        K.base_flags = (2 if sw else 0) | fpass
        if getattr(d, "_fpass", 0) != fpass:
            d.launch_cache.clear()
            d._fpass = fpass
        K.lib = L
        K.spmm(d, B, C)

    if args.once:
        for _ in range(3):
            run(args.once)
        torch.cuda.synchronize()
        return
    ref = None
    times = {k: [] for k in variants}
    for name in variants:
        run(name); run(name)
        if args.check:
            if ref is None:
                ref = C.clone()
            else:
                print(name, "max rel diff vs first:", float((C - ref).abs().max() / ref.abs().max()))
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for name in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(name); e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1))
    split = {}
    if args.split:
        for name in variants:
            d, sw, L, fp = variants[name]
            tl = TimingLib(L)
            variants[name] = (d, sw, tl, fp)
            d.launch_cache.clear()
            for _ in range(5):
                run(name)
            torch.cuda.synchronize()
            split[name] = tl.summary()
            variants[name] = (d, sw, L, fp)
            d.launch_cache.clear()
    out = {}
    for name, ts in times.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        d = variants[name][0]
        out[name] = {"median_ms": med, "min_ms": mn, "alg_GBs": alg / med / 1e6, "gather_TBs": 4 * f * nnz / med / 1e9,
                     "ntasks": d.ntasks, "nslots": d.nslots_total,
                     "core_nnz": d.core.nnz if d.core else 0, "core_pieces": d.core.npieces if d.core else 0,
                     "dense_nnz": d.dense3.nnz if d.dense3 else 0, "gather_nnz": int(d.col.numel()),
                     "strip_nnz": d.strip.nnz if d.strip else 0, "strip_pieces": d.strip.npieces if d.strip else 0,
                     "strip_recs": int(d.strip.rec.shape[0]) if d.strip else 0,
                     "split_ms": split.get(name)}
        if split.get(name):
            print("    split:", "  ".join("%s %.3f" % (k.replace("pgcn_spmm_", ""), v) for k, v in split[name].items()),
                  " gather_nnz %d dense_nnz %d strip_nnz %d pieces %d recs %d" % (
                      d.col.numel(), d.dense3.nnz if d.dense3 else 0, d.strip.nnz if d.strip else 0,
                      d.strip.npieces if d.strip else 0, d.strip.rec.shape[0] if d.strip else 0))
        print("%-14s median %.3f ms  min %.3f ms  alg %.0f GB/s (%.2f%% of 8 TB/s)  gather %.1f TB/s  tasks %d core %.1f%% pieces %d"
              % (name, med, mn, alg / med / 1e6, 100 * alg / med / 1e6 / 8000, 4 * f * nnz / med / 1e9, d.ntasks,
                 100.0 * (d.core.nnz if d.core else 0) / nnz, d.core.npieces if d.core else 0))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "spmm_probe.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
