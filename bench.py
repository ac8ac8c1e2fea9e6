#!/usr/bin/env python3
"""Benchmark of the GCN aggregation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, or self-launched)

Default workload (BASELINE.json configs[1]): Reddit-shaped graph (n = 232 965, 114 615 892 directed edges + self
loops, seeded R-MAT stand-in -- the dataset is not in the image), 3-layer GCN, f = 128, random 1D partition over N
GPUs.  One *step* = one training epoch of GPU/PGCN.py:212-220 (forward, loss, backward, gradient all-reduce, Adam)
= 2.L aggregations A.H through the HIP engine.  Metric: edges aggregated per second = 2.L.nnz / t_epoch, whole job;
ms_per_step = ms/epoch.

Inputs (the other BASELINE configs, same JSON contract):
  --workload products --generator sbm --partvec tests/golden/partvec/products-sbm.A.mtx.8.hp.gz   (config 3)
  --shards PREFIX [--partvec FILE]      every rank reads ONLY PREFIX.<rank>.pgcsr (tools/make_shards.py): the
                                        papers100M-scale path (config 4), no global matrix on any rank
  --workload reddit-gat                 3 x PGAT, 4 heads x 64 (config 5)
  --mtx A.mtx [--partvec FILE]          a real MatrixMarket file (also picked up automatically from
                                        $PGCN_DATA_DIR/<workload>.mtx, default ./data/)
  --emulate-rank r/P                    ONE GPU runs rank r of a P-rank job (exchange = no-op on resident slabs):
                                        the shard shapes that 2/4/8 GPUs run, with a roofline object per launch group

Extra objects in the JSON line:
  roofline     dominant kernel = the local-block SpMM A_loc.H, ONE launch group: spmm_tasks_kernel (XCD-sliced
               gather part) + spmm_strip_kernel (512 x 128 LDS-staged strip tiles) + spmm_dense_kernel (fp32-MFMA
               tiles) (+ spmm_core_kernel, the 128 x 128 LDS core, on small blocks) + the fix-up that adds their
               partial sums.  achieved = algorithmic bytes (8.nnz + 8.(n_r+1) + 4.f.n_c + 4.f.n_r) / the
               group's average duration, measured live with HIP events on the launch stream inside the timed
               region; peak 8 TB/s.  `traffic` comes from separate rocprofv3 --pmc passes
               (profiles/pmc_traffic.json, keyed by workload / generator / ranks, stamped with the kernel sources)
               when present, else null.
  cpu_baseline the CPU oracle (GraphBLAS-free restatement of Parallel-GCN's SpMM and training loop, OpenMP)
               timed on this host on a bounded sample: rank 0, N = 1 only.
"""
import time as _time
PROCESS_T0 = _time.time()          # (before the heavy imports: a cold box pages torch in for a long time)
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md (6.29e12 measured copy ceiling)


def pkg(sub):
    return importlib.import_module(PKG + "." + sub)


class KernelTimer:
    """Wraps HipKernels.spmm: HIP events on the launch stream around each launch of the
    dominant kernel (the local-block SpMM), tagged by operand."""

    def __init__(self, kernels, device):
        self.k, self.device, self.records, self.on = kernels, device, [], False
        self._spmm = kernels.spmm
        kernels.spmm = self.spmm

    def spmm(self, A, B, C, accumulate=False):
        if not self.on:
            return self._spmm(A, B, C, accumulate)
        s = torch.cuda.current_stream(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        out = self._spmm(A, B, C, accumulate)
        e1.record(s)
        self.records.append((id(A), B.shape[1], e0, e1))
        return out

    def summary(self, key_id, f):
        ts = [e0.elapsed_time(e1) for (i, ff, e0, e1) in self.records if i == key_id and ff == f]
        return (sum(ts) / len(ts), len(ts)) if ts else (None, 0)


def exchange_summary(probe, P, dev, world):
    """The rank-0 view of engine.ExchangeProbe.summary() plus, per (direction, round), the MAX over ranks of the transfer and of the
    exposed wait (the rank that waits longest sets the step).  Every rank calls this (collective)."""
    try:
        rep = probe.summary()
    except Exception as e:           # (first contact with real xGMI: the diagnostics must never cost the line itself)
        rep = {"error": repr(e)[:300]}
    keys = [(tag, i) for tag in ("forward", "backward") for i in range(len(rep.get(tag, [])))]
    vals = [rep[tag][i][k] for tag, i in keys for k in ("ms", "exposed_ms")]
    ar = rep.get("allreduce")
    vals += [ar["ms"], ar["exposed_ms"]] if ar else [0.0, 0.0]
    nkeys = torch.tensor([len(vals)], dtype=torch.int64, device=dev)
    if world > 1:                    # (every rank has the same rounds; a rank whose summary failed must not desynchronise the collective)
        lo = nkeys.clone()
        P._all_reduce(nkeys, dist.ReduceOp.MAX)
        P._all_reduce(lo, dist.ReduceOp.MIN)
        if int(nkeys) != int(lo):
            rep["what"] = "the ranks disagree about the exchange records (%d..%d values): no max over ranks" % (int(lo), int(nkeys))
            return rep
    t = torch.tensor(vals, dtype=torch.float64, device=dev)
    if world > 1:
        P._all_reduce(t, dist.ReduceOp.MAX)
    t = t.tolist()
    for j, (tag, i) in enumerate(keys):
        rep[tag][i]["ms_max_over_ranks"], rep[tag][i]["exposed_ms_max_over_ranks"] = t[2 * j], t[2 * j + 1]
    if ar:
        ar["ms_max_over_ranks"], ar["exposed_ms_max_over_ranks"] = t[-2], t[-1]
    rep["what"] = ("per round of the boundary all-to-all-v of ONE aggregation (mean over the timed steps, this rank): bytes sent / received, "
                   "the largest single peer segment (what one xGMI link carries), ms on the stream that runs it, exposed_ms = how long the "
                   "compute stream stood waiting for it, GB/s on that link against 153 GB/s; *_max_over_ranks: the slowest rank")
    return rep


def kernel_source_stamp():
    """sha256 over the kernel sources and the layout code: what a PMC traffic figure is valid for."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, PKG)
    files = sorted(glob.glob(os.path.join(base, "csrc", "*"))) + sorted(glob.glob(os.path.join(base, "gemm", "*"))) + \
        [os.path.join(base, x) for x in ("partition.py", "kernels.py", "tuning.py")]     # (tuning.py: the defaults the layout is built with)
    for fn in files:
        if os.path.isfile(fn) and not fn.endswith((".o", ".so")):
            h.update(os.path.basename(fn).encode())
            with open(fn, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


class SplitTimer:
    """Proxy of the C-ABI library: HIP events around every pgcn_spmm_* launch of one SpMM group, for the
    per-kernel split of the roofline object (run AFTER the timed region)."""

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

    def summary_us(self):
        acc = {}
        for name, e0, e1 in self.rec:
            acc.setdefault(name.replace("pgcn_spmm_", "").replace("_f32", ""), []).append(1e3 * e0.elapsed_time(e1))
        return {k: float(np.median(v)) for k, v in acc.items()}


def cpu_baseline_epoch(rp, ci, va, n, f, budget_s):
    """The training loop of Parallel-GCN/main.c:166-454 (oracle_pargcn_train: L-1 sigmoid GCN layers f..f,2 as
    written by the reference's preprocess for `-l 3`, BCE, SGD; 3 epochs like main.c:231) on the host cores."""
    import scipy.sparse as sp
    from oracle import oracle
    A = sp.csr_matrix((va, ci, rp), shape=(n, n))
    d = [n, f, f, 2]
    rng = np.random.default_rng(0)
    W = {l: ((rng.random((d[l], d[l + 1]), dtype=np.float32) * 2 - 1) * np.float32(np.sqrt(6.0 / (d[l] + d[l + 1]))))
         for l in (1, 2)}
    H0 = np.ones((n, f), dtype=np.float32)
    Y = np.zeros((n, 2), dtype=np.float32)
    Y[:, 1] = 1.0
    Ym = np.zeros((n, 2), dtype=np.uint8)
    Ym[:, 1] = 1
    part = np.zeros(n, dtype=np.int32)
    t0 = time.time()
    oracle.pargcn_train(A, part, 1, d, W, H0, Y, Ym, epochs=1)
    t1 = time.time() - t0
    epochs, t_total = 1, t1
    if 3 * t1 < budget_s:                       # the reference's own 3 timed epochs fit the budget
        t0 = time.time()
        oracle.pargcn_train(A, part, 1, d, W, H0, Y, Ym, epochs=3)
        t_total, epochs = time.time() - t0, 3
    spmm_per_epoch = 4                          # (L-1) = 2 layers, forward + backward aggregation each
    return {"ms_per_epoch": 1e3 * t_total / epochs, "epochs": epochs,
            "edges_per_s": spmm_per_epoch * ci.shape[0] * epochs / t_total, "spmm_per_epoch": spmm_per_epoch,
            "loop": "Parallel-GCN/main.c GCN(): config `3 n %d %d 2` (2 GCN layers, sigmoid, BCE, SGD 0.01), P = 1" % (f, f)}


def cpu_baseline_same_epoch(part, rp, ci, va, n, f, L, budget_s):
    """The epoch the GPU line times -- L x (A.H, .W^T, relu), log_softmax, nll, backward with A^T, Adam: GPU/PGCN.py:212-220
    -- in fp32 on the host cores (oracle.pgcn_epochs_f32: OpenMP SpMM of oracle/pgcn_oracle.c + BLAS GEMMs): the CPU
    figure that is comparable with the line's `value` (same workload, same 2 L aggregations per epoch)."""
    from oracle import oracle
    rpt, cit, vat = pkg("partition").full_csr(part.A_loc_T)
    csr_t = (rpt.cpu().numpy().astype(np.int64), cit.cpu().numpy().astype(np.int32), vat.cpu().numpy().astype(np.float32))
    torch.manual_seed(0)
    W = [torch.nn.Linear(f, f, bias=False).weight.detach().numpy() for _ in range(L)]
    own = part.owned.cpu().numpy()
    H0 = np.repeat(own.astype(np.float32)[:, None], f, axis=1)            # PGCN.py:187-189, local row i = global row owned[i]
    labels = own % f
    _, secs = oracle.pgcn_epochs_f32((rp, ci, va), csr_t, W, H0, labels, 1)           # first epoch: page faults, thread start
    epochs = int(max(1, min(4, budget_s // max(secs[0], 1e-3))))
    _, secs2 = oracle.pgcn_epochs_f32((rp, ci, va), csr_t, W, H0, labels, epochs)
    t = float(np.mean(secs2))
    return {"ms_per_epoch": 1e3 * t, "epochs": epochs, "edges_per_s": 2 * L * ci.shape[0] / t, "spmm_per_epoch": 2 * L,
            "loop": "GPU/PGCN.py run(): %d x (A.H, .W^T, relu), log_softmax, nll, backward, Adam -- the epoch of this line's `value`" % L}


def cpu_reference_binary(budget_s=30.0):
    """The reference's OWN CPU engine -- /root/reference/Parallel-GCN/main.c compiled unmodified (oracle/_ref/grbgcn, built
    in the build container by `make -C oracle ref`, shipped prebuilt with the snapshot) -- timed by its own `time : %f secs`
    line (3 epochs, main.c:229-445) on the `mid` workload (n = 131 072, 4.3 M entries, f = 64, 2 layers: what its text
    parser reads in seconds).  Its GraphBLAS / MPI are the single-threaded stand-ins of oracle/shim (SuiteSparse and MPI
    are absent): kind "reference-binary-on-standins", 1 thread.  None when the binary is not there."""
    import re
    import subprocess
    import tempfile
    import scipy.sparse as sp
    binary = os.path.join(ROOT, "oracle", "_ref", "grbgcn")
    if not os.path.exists(binary):
        return None
    synth, io_ = pkg("synth"), pkg("pargcn_io")
    n, _, f, L = synth.SHAPES["mid"]
    n, row, col, val = synth.make_graph("mid", seed=0)
    A = sp.csr_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n))
    with tempfile.TemporaryDirectory() as tmp:
        io_.write_directory(tmp, A, np.zeros(n, np.int64), 1, L, f, value_format="%.9g")
        env = dict(os.environ, MPISHIM_NP="1", MPISHIM_SEED="1")
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "oracle", "_build") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
        res = subprocess.run([binary, "-p", tmp, "-c", os.path.join(tmp, "config"), "-t", "1"], env=env, capture_output=True,
                             text=True, timeout=max(60.0, 4 * budget_s))
    if res.returncode != 0:
        return {"error": res.stderr[-300:]}
    secs = float(re.search(r"time : ([0-9.]+) secs", res.stdout).group(1))
    return {"kind": "reference-binary-on-standins", "workload": "mid (n=131072, nnz=%d, f=%d, %d layers)" % (A.nnz, f, L),
            "threads": 1, "ms_per_epoch": 1e3 * secs / 3, "edges_per_s": 2 * (L - 1) * A.nnz / (secs / 3),
            "aggregations_per_epoch": 2 * (L - 1),
            "what": "Parallel-GCN/main.c unmodified (oracle/_ref/grbgcn), its own `time :` line over 3 epochs; GraphBLAS / MPI = the "
                    "single-threaded stand-ins of oracle/shim, NOT SuiteSparse"}


def cpu_baseline(part, f, L=3, budget_s=12.0):
    """The CPU path timed beside the GPU kernel, on a bounded sample (about 30 s of host time in total):
      value  = the SAME epoch as the GPU line (GPU/PGCN.py's loop, fp32, all host cores) -> edges aggregated per second;
      spmm   = the oracle's OpenMP CSR SpMM alone on the full local block (1 of the 2 L per epoch);
      pargcn_epoch = 3 epochs of the restated Parallel-GCN/main.c loop (another model: 2 sigmoid layers, BCE, SGD);
      reference = the reference's own binary on the `mid` workload (when oracle/_ref/grbgcn is shipped)."""
    from oracle import oracle
    Lb = oracle.lib()
    rp, ci, va = pkg("partition").full_csr(part.A_loc)     # whole local block, core entries included
    rp = rp.cpu().numpy().astype(np.int64)
    ci = ci.cpu().numpy().astype(np.int32)
    va = va.cpu().numpy().astype(np.float32)
    n = rp.shape[0] - 1
    rng = np.random.default_rng(0)
    B = rng.random((n, f), dtype=np.float32)
    C = np.empty((n, f), dtype=np.float32)
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    reps, t_total = 0, 0.0
    while True:
        t0 = time.time()
        Lb.oracle_spmm_csr_f32(n, p(rp, ctypes.c_int64), p(ci, ctypes.c_int32), p(va, ctypes.c_float),
                               p(B, ctypes.c_float), f, p(C, ctypes.c_float), f, f, 0)
        t_total += time.time() - t0
        reps += 1
        if t_total + t_total / reps > budget_s / 4 or reps >= 4:
            break
    cores = Lb.oracle_num_threads()
    out = {"value": None, "unit": "edges aggregated/s", "cores": cores, "host_cpus": os.cpu_count(), "kind": "port",
           "spmm": {"edges_per_s": ci.shape[0] * reps / t_total, "ms_per_spmm": 1e3 * t_total / reps, "reps": reps}}
    del B, C
    same = cpu_baseline_same_epoch(part, rp, ci, va, n, f, L, budget_s)
    out["value"] = same["edges_per_s"]
    out["same_epoch_as_gpu"] = same
    out["sample"] = ("%d epoch(s) of the line's own workload (full graph, nnz=%d, f=%d, %d aggregations per epoch) on %d threads: "
                     "%.0f ms/epoch; oracle/pgcn_oracle.c OpenMP SpMM (%.0f ms each) + BLAS GEMMs, fp32"
                     % (same["epochs"], ci.shape[0], f, same["spmm_per_epoch"], cores, same["ms_per_epoch"], out["spmm"]["ms_per_spmm"]))
    try:
        out["pargcn_epoch"] = cpu_baseline_epoch(rp, ci, va, n, f, budget_s)
    except Exception as e:
        out["pargcn_epoch"] = {"error": repr(e)}
    try:
        ref = cpu_reference_binary()
        if ref is not None:
            out["reference"] = ref
    except Exception as e:
        out["reference"] = {"error": repr(e)}
    return out


class NoExchange:
    """--emulate-rank: the exchange of an N-rank job replaced by a no-op on slabs that already hold rows (one GPU
    measures the COMPUTE side of rank r; nothing is sent)."""
    name = "none (emulated rank)"

    def alltoallv(self, *a):
        pass

    def allreduce_sum(self, buf):
        pass

    def close(self):
        pass


class PacedExchange(NoExchange):
    """--emulate-rank r/P --pace-exchange GBs: nothing is sent, but every round OCCUPIES the stream it is issued on (the comm
    stream) for as long as its largest peer segment would take on one xGMI link at the given rate (a spin kernel: no memory
    traffic, one wave), the gradient all-reduce for a fixed 25 us.  With engine.ExchangeProbe on, the line then says how much of a
    transfer of that length the aggregation hides (`exposed_ms`) -- a PRICE of the N > 1 overlap from one GPU, not a measurement
    of xGMI (r06)."""

    def __init__(self, rank, gbs, dev):
        self.rank, self.rate, self.dev = rank, float(gbs) * 1e9, dev
        self.name = "none (emulated rank; every round paced at %.0f GB/s per link)" % gbs
        # cycles of torch.cuda._sleep per microsecond, measured once
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000000)
        e0.record()
        torch.cuda._sleep(20000000)
        e1.record()
        torch.cuda.synchronize(dev)
        self.cycles_per_us = 20000000.0 / max(e0.elapsed_time(e1) * 1e3, 1e-3)

    def _spin(self, seconds):
        c = int(seconds * 1e6 * self.cycles_per_us)
        if c > 0:
            torch.cuda._sleep(c)

    def alltoallv(self, send, send_off, recv, recv_off, f):
        size = len(send_off) - 1
        peer = max([max(send_off[q + 1] - send_off[q], recv_off[q + 1] - recv_off[q]) for q in range(size) if q != self.rank] or [0])
        self._spin(peer * f * 4 / self.rate)

    def allreduce_sum(self, buf):
        self._spin(25e-6)


def real_mtx_for(workload):
    """A real dataset dropped next to the benchmark is used instead of the synthetic stand-in:
    $PGCN_DATA_DIR/<workload>.mtx (default ./data/), written e.g. by the reference's preprocess/GrB-GNN-IDG.py."""
    d = os.environ.get("PGCN_DATA_DIR", os.path.join(ROOT, "data"))
    for name in (workload + ".mtx", workload + ".A.mtx"):
        path = os.path.join(d, name)
        if os.path.exists(path):
            return path
    return None


def read_partvec_arg(args, n, parts, synth, partition):
    """(part vector tensor, description) for `parts` ranks from --partvec (random | block | FILE)."""
    if parts == 1:
        return torch.zeros(n, dtype=torch.int64), "none"
    if args.partvec == "random":
        return synth.random_partvec(n, parts, seed=0), "random (seeded, GCN-HP/main.cpp:133-142)"
    if args.partvec == "block":
        return synth.block_partvec(n, parts), "contiguous blocks"
    pv = partition.read_partvec(args.partvec)
    if len(pv) != n or max(pv) >= parts or min(pv) < 0:
        sys.exit("part vector %s: %d entries / parts 0..%d, graph has %d vertices on %d ranks"
                 % (args.partvec, len(pv), max(pv), n, parts))
    return torch.tensor(pv, dtype=torch.int64), "file:" + os.path.basename(args.partvec)


def acquire_partition(args, rank, world, dev, stage, with_transpose=True):
    """Rank `rank`'s Partition from one of the three input kinds.  Returns (part, info): info = {n, nnz,
    partition, source, data}.  With --emulate-rank r/P (world == 1) the partition of rank r of P is built."""
    synth, partition, ingest = pkg("synth"), pkg("partition"), pkg("ingest")
    prank, parts = rank, world
    if args.emulate_rank:
        if world != 1:
            sys.exit("--emulate-rank runs on one GPU (--gpus 1)")
        prank, parts = [int(x) for x in args.emulate_rank.split("/")]
        if not 0 <= prank < parts:
            sys.exit("--emulate-rank r/P needs 0 <= r < P")
    mtx = args.mtx or (None if args.shards else real_mtx_for(args.workload.replace("-gat", "")))
    if args.shards:
        # ---- binary CSR shards: this rank reads ONLY its own rows (papers100M-scale path) ---------------
        path = ingest.shard_path(args.shards, prank)
        sh = ingest.read_shard(path)
        n = int(sh["n"])
        if args.partvec in ("random", "block"):
            cand = "%s.%d.bp" % (args.shards, parts)          # tools/make_shards.py writes the block vector it used
            if os.path.exists(cand):
                args.partvec = cand
            elif os.path.exists(args.shards + ".degree.npy"):
                args.partvec = "block"                        # --only-rank shards: contiguous blocks, computed (no 111 M-id text file)
            else:
                sys.exit("--shards needs the part vector the shards were cut with (--partvec FILE or %s)" % cand)
        partvec, pname = read_partvec_arg(args, n, parts, synth, partition)
        if sh["nparts"] != parts or sh["rank"] != prank:
            sys.exit("%s was written for rank %d of %d, this is rank %d of %d" % (path, sh["rank"], sh["nparts"], prank, parts))
        if not np.array_equal(sh["rows"], np.nonzero(partvec.numpy() == prank)[0]):
            sys.exit("%s does not hold the rows the part vector gives rank %d" % (path, prank))
        r_, c_, v_ = ingest.shard_coo(sh)
        row, col, val = torch.from_numpy(r_).to(dev), torch.from_numpy(c_).to(dev), torch.from_numpy(v_).to(dev)
        del sh, r_, c_, v_
        stage("shard read")
        if world > 1:
            part = partition.build_partition_local(row, col, val, n, partvec, prank, parts, with_transpose=with_transpose)
        elif parts > 1:
            # one process stands in for rank prank: what the two collectives of build_partition_local would have brought comes
            # from the side files of tools/make_shards.py --only-rank (global degree vector; symmetric pattern)
            dfile = args.shards + ".degree.npy"
            if not os.path.exists(dfile):
                sys.exit("--emulate-rank with --shards needs %s (tools/make_shards.py --only-rank)" % dfile)
            deg = torch.from_numpy(np.load(dfile).astype(np.int64)).to(dev)
            meta = {}
            if os.path.exists(args.shards + ".meta.json"):
                with open(args.shards + ".meta.json") as fh:
                    meta = json.load(fh)
            part = partition.build_partition_local(row, col, val, n, partvec, prank, parts, with_transpose=with_transpose,
                                                   emulate={"gdeg": 2 * deg, "nnz_global": meta.get("nnz_global", int(row.numel()))})
            del deg
        else:
            part = partition.build_partition(row, col, val, n, partvec, 0, 1, with_transpose=with_transpose)
        info = {"n": n, "nnz": int(part.nnz_global), "partition": pname, "data": "synthetic (shards)" if not args.real else "real",
                "source": "binary CSR shards %s.<rank>.pgcsr (rank-local ingest, no global matrix)" % os.path.basename(args.shards)}
        return part, info
    if mtx:
        # ---- a MatrixMarket file (real dataset when present) ---------------------------------------------
        n = int(ingest.mtx_info(mtx)["nrows"])
        partvec, pname = read_partvec_arg(args, n, parts, synth, partition)
        if world > 1:
            A = ingest.load_partition(mtx, partvec.numpy(), prank)       # this rank's rows only (C++ reader)
        else:
            A = ingest.mmread(mtx)
        row, col = torch.from_numpy(A.row.astype(np.int64)).to(dev), torch.from_numpy(A.col.astype(np.int64)).to(dev)
        val = torch.from_numpy(A.data.astype(np.float32)).to(dev)
        del A
        stage("mtx read")
        if world > 1:
            part = partition.build_partition_local(row, col, val, n, partvec, prank, parts, with_transpose=with_transpose)
        else:
            part = partition.build_partition(row, col, val, n, partvec, prank, parts, with_transpose=with_transpose)
        info = {"n": n, "nnz": int(part.nnz_global), "partition": pname, "data": "file",
                "source": "MatrixMarket file %s" % os.path.basename(mtx)}
        return part, info
    # ---- synthetic graph (same seed on every rank) ----------------------------------------------------------
    base = args.workload.replace("-gat", "")
    n, row, col, val = synth.make_graph(base, seed=0, device=dev, generator=args.generator)
    nnz = int(row.numel())
    partvec, pname = read_partvec_arg(args, n, parts, synth, partition)
    if world > 1:   # all ranks must hold the same graph
        chk = torch.stack([row.sum(), col.sum(), (val.double().sum() * 1e6).long()]).double()
        lo, hi = chk.clone(), chk.clone()
        P = pkg("PGCN")
        P._all_reduce(lo, dist.ReduceOp.MIN)
        P._all_reduce(hi, dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks generated different graphs"
    stage("graph ready")
    part = partition.build_partition(row, col, val, n, partvec, prank, parts, with_transpose=with_transpose)
    info = {"n": n, "nnz": nnz, "partition": pname, "data": "synthetic",
            "source": "%s-like %s" % (base, "R-MAT" if args.generator == "rmat" else "planted-partition (SBM)")}
    return part, info


def pmc_traffic_for(args, world, f, block="loc"):
    """(bytes or None, note): L2<->fabric bytes of the dominant launch group from the committed rocprofv3 --pmc
    passes (profiles/pmc_traffic.json: one record per (workload, generator, ranks, f)), valid only for the kernel
    sources they were taken on (sha256 stamp)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc):
        return None, "no PMC record"
    try:
        with open(pmc) as fh:
            doc = json.load(fh)
    except Exception:
        return None, "unreadable PMC record"
    recs = doc["records"] if isinstance(doc, dict) and "records" in doc else [doc]
    ranks = args.emulate_rank or str(world)
    note = "no PMC record for this workload"
    for rec in recs:
        same = (rec.get("workload") == args.workload and str(rec.get("ranks", rec.get("n_gpus"))) == ranks
                and rec.get("f") == f and rec.get("generator", "rmat") == args.generator
                and rec.get("partvec", "random") == os.path.basename(args.partvec) and rec.get("block", "loc") == block)
        if not same:
            continue
        if rec.get("source_stamp") == kernel_source_stamp():
            return rec.get("hbm_bytes_per_launch"), ("rocprofv3 --pmc passes of %s (profiles/pmc_traffic.json), same kernel sources"
                                                     % rec.get("source", "?"))
        note = "PMC record is stale (kernel sources changed since it was taken): dropped"
    return None, note


def multirank_selftest(rank, world, dev, kernels, exch, n=8192, nnz=400000, f=32):
    """First-contact check of an N-rank job, BEFORE anything is timed: a small synthetic graph (same seed on every rank)
    is cut with a random part vector, every rank runs the aggregation forward and backward through its real engine and
    the real boundary exchange (two rounds, non-empty segments to every peer), and rank 0 compares the gathered rows
    with the SAME kernels run on the whole graph as one rank (no exchange): what differs is exactly the multi-rank
    path -- packed rows, slab order, halo products, the reverse exchange with accumulation.  Every wait has a
    deadline (engine._wait_or_die).  Returns a record for the JSON line; raises on a mismatch (every rank)."""
    synth, partition, engine = pkg("synth"), pkg("partition"), pkg("engine")
    cpu = dev.type != "cuda"
    n, row, col, val = synth.make_graph(n, nnz, seed=11, device="cpu")
    pv = synth.random_partvec(n, world, seed=5)
    part = partition.build_partition(row, col, val, n, pv, rank, world)
    eng = engine.AggregationEngine(part, kernels, dev, exch)
    g = torch.Generator()
    g.manual_seed(77)
    Hfull = torch.rand(n, f, generator=g) * 2 - 1
    Gfull = torch.rand(n, f, generator=g) * 2 - 1
    own = part.owned
    fwd = eng.forward(Hfull[own].to(dev))
    bwd = eng.backward(Gfull[own].to(dev))
    if not cpu:
        engine._wait_or_die(dev, "the %d-rank forward / backward aggregation of the self-test" % world)
    mine = {"own": own.numpy(), "fwd": fwd.cpu().numpy(), "bwd": bwd.cpu().numpy()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    rec = {"ranks": world, "n": n, "nnz": int(row.numel()), "f": f, "rounds": eng.rounds, "exchange": getattr(exch, "name", "?"),
           "exchanger_selftest": getattr(exch, "selftest", None), "boundary_rows_this_rank": int(part.n_send)}
    verdict = [1.0, 0.0, 0.0]
    if rank == 0:
        one = partition.build_partition(row, col, val, n, torch.zeros(n, dtype=torch.int64), 0, 1)
        e1 = engine.AggregationEngine(one, kernels, dev, None)
        o1 = one.owned
        ref_f = np.zeros((n, f), np.float32)
        ref_b = np.zeros((n, f), np.float32)
        ref_f[o1.numpy()] = e1.forward(Hfull[o1].to(dev)).cpu().numpy()
        ref_b[o1.numpy()] = e1.backward(Gfull[o1].to(dev)).cpu().numpy()
        got_f, got_b = np.full((n, f), np.nan, np.float32), np.full((n, f), np.nan, np.float32)
        for r in everyone:
            got_f[r["own"]], got_b[r["own"]] = r["fwd"], r["bwd"]
        ef = float(np.abs(got_f - ref_f).max() / max(np.abs(ref_f).max(), 1e-30))
        eb = float(np.abs(got_b - ref_b).max() / max(np.abs(ref_b).max(), 1e-30))
        verdict = [1.0 if (ef < 1e-5 and eb < 1e-5) else 0.0, ef, eb]        # (NaN = a row nobody delivered: fails)
    box = [verdict]
    dist.broadcast_object_list(box, src=0)
    rec["forward_rel_err_vs_one_rank"], rec["backward_rel_err_vs_one_rank"] = box[0][1], box[0][2]
    if box[0][0] != 1.0:
        raise RuntimeError("multi-rank self-test FAILED: %s" % json.dumps(rec))
    return rec


def graph_replay(model, make_model, H, labels, n, P, steps, dev, eager_ms, world=1, on_timeout=None):
    """Capture ONE training step (forward, loss, backward, gradient all-reduce, Adam) in a HIP graph and time `steps`
    replays.  The C-ABI library never allocates or synchronises and the engine orders its streams with events only, so
    the whole step -- ~80 launches on a whole graph, ~150 on a shard with its halo groups, the RCCL calls of the
    exchange and of average_gradients included (RCCL records grouped send / recv and all-reduce into a capturing
    stream) -- is capturable; a replay has no host-side enqueue cost.  Eager timing stays the headline `value` (it
    carries the live per-kernel events).  N > 1: every rank must have captured, or nobody replays; the replays are
    waited for with a deadline."""
    engine = pkg("engine")

    voted = [False]

    def all_agree(ok):
        voted[0] = True
        if world == 1:
            return ok
        t = torch.tensor([1.0 if ok else 0.0], device=dev)
        P._all_reduce(t, dist.ReduceOp.MIN)
        return float(t) == 1.0
    try:
        # fresh leaves: the AccumulateGrad nodes of the eager run's parameters are bound to the default stream, which
        # must not be touched during a capture
        fresh = make_model()
        fresh.load_state_dict(model.state_dict())
        model = fresh
        H = H.detach().clone().requires_grad_(True)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # warm-up of the capturable optimizer off the default stream
            for _ in range(2):
                opt.zero_grad(set_to_none=True)
                P.local_loss(model(H), labels, n).backward()
                opt.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        err = None
        try:
            with torch.cuda.graph(g):
                loss = P.local_loss(model(H), labels, n)
                loss.backward()
                P.average_gradients(model)
                opt.step()
        except Exception as e:
            err = repr(e)[:300]
        if not all_agree(err is None):
            return {"captured": False, "error": err or "another rank could not capture"}
        torch.cuda.synchronize(dev)
        for _ in range(2):
            g.replay()
        engine._wait_or_die(dev, "the first replays of the captured %d-rank training step" % world, on_timeout=on_timeout)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        t_host = time.perf_counter() - t0
        if world > 1:                                      # (a deadline instead of a bare synchronize: 0.2 ms granularity, N > 1 only)
            engine._wait_or_die(dev, "the %d timed replays of the captured %d-rank training step" % (steps, world),
                                on_timeout=on_timeout, poll_s=0.0002)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t_all = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([t_all], dtype=torch.float64, device=dev)
            P._all_reduce(t, dist.ReduceOp.MAX)
            t_all = float(t)
        return {"captured": True, "ms_per_step": 1e3 * t_all / steps, "host_enqueue_ms_per_step": 1e3 * t_host / steps,
                "eager_ms_per_step": eager_ms, "loss": float(loss), "ranks": world}
    except Exception as e:                                 # a capture problem must not cost the eager line
        if not voted[0]:                                   # (the others wait in the vote)
            all_agree(False)
        return {"captured": False, "error": repr(e)[:300]}


class GatKernelTimer:
    """HIP events around every launch of the dominant GAT kernel: the fused transposed product + edge gradient
    (pgcn_spmm_heads_grad_f32), or -- shapes it does not cover -- the stand-alone edge gradient."""
    NAMES = ("spmm_heads_grad", "gat_edge_grad_tasks", "gat_edge_grad_sliced")
    PARTS = ("gat_blocks_backward", "gat_blocks_forward", "spmm_heads_forward2")   # r06: the block part of the same pass, and the forward pass

    def __init__(self, kernels, device):
        self.k, self.device, self.records, self.on = kernels, device, {}, False
        for name in self.NAMES + self.PARTS:                                # whichever variant the engine uses
            if hasattr(kernels, name):
                setattr(kernels, name, self._wrap(name, getattr(kernels, name)))

    def _wrap(self, name, fn):
        def call(*a, **kw):
            if not self.on:
                return fn(*a, **kw)
            s = torch.cuda.current_stream(self.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            out = fn(*a, **kw)
            e1.record(s)
            if out:
                self.records.setdefault(name, []).append((e0, e1))
            return out
        return call

    def mean_ms(self):
        """(kernel name, mean launch ms, launches) of the variant that ran."""
        for name in self.NAMES:
            ts = [a.elapsed_time(b) for a, b in self.records.get(name, [])]
            if ts:
                return name, sum(ts) / len(ts), len(ts)
        return None, None, 0

    def part_ms(self, name):
        ts = [a.elapsed_time(b) for a, b in self.records.get(name, [])]
        return sum(ts) / len(ts) if ts else None


def bench_gat(args, rank, world, dev, backend, stage):
    """BASELINE config 5: Reddit-shaped 3-layer GAT, 4 heads x 64 (the reference's PGAT.py layer on the stored
    entries: edge softmax + multi-head weighted SpMM + their backward).  One step = one epoch of GPU/PGAT.py:
    205-221 (forward, loss, backward, gradient all-reduce, Adam)."""
    synth, partition, engine, kernels = pkg("synth"), pkg("partition"), pkg("engine"), pkg("kernels")
    G, gat = pkg("PGAT"), pkg("gat")
    base = args.workload[:-4]
    heads, dh, L = args.heads, args.features or 64, args.layers or 3
    F = heads * dh
    t0 = time.time()
    part, info = acquire_partition(args, rank, world, dev, stage, with_transpose=False)
    n, nnz = info["n"], info["nnz"]
    K = kernels.HipKernels(dev)
    emul = bool(args.emulate_rank and part.size > 1)
    if emul:
        exch = PacedExchange(part.rank, args.pace_exchange, dev) if args.pace_exchange > 0 else NoExchange()
    else:
        exch = engine.make_exchanger(rank, world, dev, os.environ.get("PGCN_EXCHANGE", "auto")) if world > 1 else None
    eng = gat.GatEngine(part, K, dev, exch, mode="standard")
    G.device, G.myrank, G.world_size, G.heads, G._engine_current = dev, rank, world, heads, eng
    pkg("PGCN").tune_dense_gemms(part.n_local, F, dev, fout=F + 2 * heads)   # library GEMM choice for H.[W^T | W^T a2 | W^T a1] made in set-up
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    stage("gat engine ready")
    torch.manual_seed(0)
    model = nn.Sequential(*[G.PGAT(eng, F, F, heads=heads) for _ in range(L)]).to(dev)
    G.initiliaze_parameters(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    H = torch.rand(part.n_local, F, device=dev, generator=gen).requires_grad_(True)
    labels = part.owned.to(dev) % F

    def step():
        logits = model(H)
        loss = G.local_loss(logits, labels, n)
        opt.zero_grad()
        loss.backward()
        G.sum_gradients(model)
        opt.step()
        return loss

    timer = GatKernelTimer(K, dev)
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.on = not args.no_kernel_timing
    xprobe = None
    paced = bool(emul and args.pace_exchange > 0)
    if (world > 1 or paced) and not args.no_kernel_timing and (paced or not emul):      # what the exchange costs, what of it is exposed (r06)
        xprobe = eng.probe = pkg("engine").ExchangeProbe(dev)
        xprobe.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.on = False
    exchange_report = None
    if xprobe is not None:
        xprobe.on = False
        exchange_report = exchange_summary(xprobe, pkg("PGCN"), dev, world)
        eng.probe = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        pkg("PGCN")._all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t)
    ms = 1e3 * elapsed / args.steps
    roofline = None
    kname, avg, launches = timer.mean_ms()
    if avg:
        n_r, n_c = part.n_local, part.n_local + part.n_halo
        if kname == "spmm_heads_grad":
            # col + de per entry and head; dOut panel gathered, Z panel and dZ rows streamed once; statistics are L2-sized
            alg = (4 + 4 * heads) * eng.nnz + 4 * F * (n_r + 2 * n_c)
            label = ("spmm_heads_kernel<%d, recompute, grad> (pgcn_spmm_heads_grad_f32: ONE gather pass over the transposed "
                     "structure = A_alpha^T . dOut + SDDMM <dOut_i, Z_j> + softmax / LeakyReLU backward + ds2)" % heads)
        else:
            alg = (4 + 8 * heads) * eng.nnz + 4 * F * (n_c + n_r)  # col + alpha + de per entry and head; Z and dOut panels
            label = ("%s_kernel (XCD-sliced SDDMM <dOut_i, Z_j> + softmax / LeakyReLU backward, one pass over the "
                     "stored entries)" % kname)
        blocks_ms = timer.part_ms("gat_blocks_backward") if kname == "spmm_heads_grad" and eng.parts else None
        gather_ms = avg
        if blocks_ms:                      # r06: the pass = the gather kernel over the remaining entries + the dense blocks on the matrix cores
            avg += blocks_ms
            label += (" + gat_blocks_kernel<backward> (pgcn_gat_blocks_backward_f32: %.0f %% of the entries in 512 x 128 blocks on the bf16 "
                      "matrix cores, weights computed in registers)" % (100.0 * eng.blocks_nnz / max(eng.nnz, 1)))
        ach = alg / (avg * 1e-3)
        traffic, traffic_note = (pmc_traffic_for(args, world, F, "gat_grad") if kname == "spmm_heads_grad" else (None, "no PMC record"))
        roofline = {"bound": "hbm", "kernel": label,
                    "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK, "traffic": traffic,
                    "traffic_note": traffic_note,
                    "alg_bytes_per_launch": alg, "avg_launch_ms": avg, "launches_timed": launches,
                    "gather_model_GBs": 4.0 * F * eng.nnz / (avg * 1e-3) / 1e9,
                    "pass_split_ms": {"backward_gather": gather_ms, "backward_blocks": blocks_ms,
                                      "forward_gather": timer.part_ms("spmm_heads_forward2"), "forward_blocks": timer.part_ms("gat_blocks_forward")}}
    out = {"metric": "edges aggregated/sec (%s-shaped %d-layer GAT, %d heads x %d, full training epoch)" % (
               base.capitalize(), L, heads, dh),
           "value": 2 * L * (eng.nnz if emul else nnz) * args.steps / elapsed, "unit": "edges/s", "n_gpus": world,
           "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": info["data"],
           "config": {"workload": "%s n=%d nnz=%d, %d-layer GAT %d heads x %d (edge softmax + multi-head weighted "
                                  "SpMM, PGAT.py layer on the stored entries), 1D partition (%s) over %d GPU(s)%s"
                                  % (info["source"], n, nnz, L, heads, dh, info["partition"], part.size,
                                     "; COMPUTE of rank %s alone on one GPU, exchange = no-op" % args.emulate_rank if emul else ""),
                      "n": n, "nnz": nnz, "heads": heads, "head_dim": dh, "layers": L, "generator": args.generator,
                      "partition": info["partition"], "emulated_rank": args.emulate_rank if emul else None,
                      "exchange": exch.name if exch else "none",
                      "rank_shape": {"n_local": part.n_local, "n_halo": part.n_halo, "n_send": part.n_send, "nnz_rank": eng.nnz},
                      "multi_head_spmm": bool(eng.multi_head), "fused_edge_gradient": bool(eng.fused_grad),
                      "blocks": None if not eng.parts else {"entries_on_blocks": eng.blocks_nnz / max(eng.nnz, 1), "structures": {
                          k: {"entries": v[1].nnz, "blocks": int(v[1].blk_img.numel()), "pieces": v[1].npieces, "panels": v[1].npanels}
                          for k, v in eng.parts.items()}},
                      "vertex_order": {k: v for k, v in (part.order_info or {}).items() if not k.startswith("_")}},
           "roofline": roofline, "ms_per_epoch": ms, "ms_per_layer_fwd_bwd": ms / L, "loss": float(loss), "setup_s": setup_s,
           "cpu_baseline": None}
    if world > 1:
        vol = torch.tensor([eng.stats["send_volume"]], dtype=torch.float64, device=dev)
        pkg("PGCN")._all_reduce(vol)
        out["exchange_rows_total"] = float(vol)
    if exchange_report is not None:
        out["exchange"] = exchange_report
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        exch.close()
        dist.destroy_process_group()


def self_launch(nproc):
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # RCCL shares device buffers between the ranks of a node through HIP IPC handles; this image's host driver
    # only supports the dmabuf flavour (the legacy mode fails with `hipIpcGetMemHandle: invalid argument`).  The
    # image exports the variable already -- this only keeps a stripped-down environment from losing it.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // nproc)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="reddit")
    ap.add_argument("--features", type=int, default=None)
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of host time per cpu_baseline leg")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--heads", type=int, default=4, help="attention heads of the *-gat workloads")
    ap.add_argument("--partvec", default="random",
                    help="random | block | a part-vector file as written by the reference's partitioners "
                         "(GPU/hypergraph/main.cpp:51-63, GPU/graph/main.cpp: one line of n part ids) for N > 1")
    ap.add_argument("--generator", default="rmat", choices=["rmat", "sbm"],
                    help="synthetic graph family: R-MAT (headline) or a planted-partition graph with a power-law tail")
    ap.add_argument("--shards", default=None, metavar="PREFIX",
                    help="binary CSR shards PREFIX.<rank>.pgcsr (tools/make_shards.py): every rank reads only its own "
                         "rows; the part vector is --partvec FILE or PREFIX.<N>.bp")
    ap.add_argument("--mtx", default=None, help="a MatrixMarket adjacency file (like PGCN.py -a) instead of the synthetic graph")
    ap.add_argument("--real", action="store_true", help="label the --shards / --mtx input as real data in the JSON line")
    ap.add_argument("--no-selftest", action="store_true",
                    help="N > 1: skip the small-graph check of the multi-rank path that runs before the timed region")
    ap.add_argument("--graph", nargs="?", const="on", default="auto", choices=("on", "off", "auto"),
                    help="after the timed (eager) region: capture ONE training step in a HIP graph -- the RCCL calls of an "
                         "N > 1 run included -- and time its replays (reported as `graph_replay`; all ranks or none).  auto "
                         "(default): on for N > 1, where the host-side enqueue of ~150 launches per step is 13 %% of a rank's "
                         "step (r04: 3.06 ms eager, 2.71 replayed on rank 0 of 8), off for N = 1 (device-bound: 10.66 vs 10.64 ms)")
    ap.add_argument("--emulate-rank", default=None, metavar="r/P",
                    help="one GPU runs rank r of a P-rank job with a no-op exchange (per-rank compute of 2/4/8 GPUs)")
    ap.add_argument("--pace-exchange", type=float, default=0.0, metavar="GBs",
                    help="with --emulate-rank: every exchange round occupies the comm stream as long as its largest peer segment takes at "
                         "this rate per link (153 = one xGMI link) -- prices the overlap, reported under `exchange`")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    wd = float(os.environ.get("PGCN_BENCH_WATCHDOG", "0"))    # debugging aid: dump every thread's stack and exit
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    t_start = time.time()

    stages = []                                   # (name, seconds since start): the set-up breakdown of the JSON line

    def stage(msg):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        stages.append((msg, round(time.time() - t_start, 3)))
        if wd > 0:
            print("[bench rank %d +%.1fs] %s" % (rank, time.time() - t_start, msg), file=sys.stderr, flush=True)
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves (one per GPU), exactly the
            # command the driver would use; rank 0's JSON line goes to our stdout.  The reference's
            # own main() spawns its ranks the same way (GPU/PGAT.py:267-275).
            sys.exit(self_launch(args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no CPU fallback in the product path) [rank %d of %d]" % (rank, world),
              file=sys.stderr, flush=True)
        time.sleep(3.0 if world > 1 else 0.0)     # let every rank say so before the launcher tears the job down
        sys.exit(1)
    dev = torch.device("cuda:%d" % (local_rank % torch.cuda.device_count()))   # (gloo dry-run: ranks share cuda:0)
    torch.cuda.set_device(dev)
    backend = os.environ.get("PGCN_BENCH_BACKEND", "nccl")   # "gloo": dry-run of the N>1 path on one GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.workload.endswith("-gat"):
        return bench_gat(args, rank, world, dev, backend, stage)
    synth, partition, engine, kernels, P = pkg("synth"), pkg("partition"), pkg("engine"), pkg("kernels"), pkg("PGCN")
    shape = synth.SHAPES.get(args.workload)
    if shape is None and not (args.shards or args.mtx):
        sys.exit("unknown workload %r (known: %s; or give --shards / --mtx)" % (args.workload, ", ".join(synth.SHAPES)))
    f = args.features or (shape[2] if shape else 128)
    L = args.layers or (shape[3] if shape else 3)

    # ---- inputs: synthetic graph / shards / MatrixMarket file -> this rank's partition ----------------
    t0 = time.time()
    part, info = acquire_partition(args, rank, world, dev, stage)
    n, nnz, partition_name = info["n"], info["nnz"], info["partition"]
    stage("partition built")
    K = kernels.HipKernels(dev)
    if args.emulate_rank and part.size > 1:
        exch = PacedExchange(part.rank, args.pace_exchange, dev) if args.pace_exchange > 0 else NoExchange()
    else:
        exch = engine.make_exchanger(rank, world, dev, os.environ.get("PGCN_EXCHANGE", "auto")) if world > 1 else None
    eng = engine.AggregationEngine(part, K, dev, exch)
    if args.emulate_rank and part.size > 1:          # the slabs a real exchange would fill: resident random rows
        eng._slab("halo", eng.n_halo, f).uniform_()
        eng._slab("send", eng.n_send, f).uniform_()
    selftest = None
    if world > 1 and not args.no_selftest:        # first contact: the multi-rank path on a small graph, before anything is timed
        selftest = multirank_selftest(rank, world, dev, K, exch)
        stage("multi-rank self-test passed")
    P._engine_current = eng           # gradient all-reduce rides the exchange's communicator and stream
    stage("engine ready (kernel structures uploaded)")
    gemm_tuned = P.tune_dense_gemms(part.n_local, f, dev)      # library GEMM choice made in set-up, not in a timed step
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    stage("GEMM choice made")

    # ---- model: L x PGCN(f, f), PGCN.py:194-200 ----------------------------------
    P.device, P.myrank, P.world_size = dev, rank, world
    P.init_stats()
    torch.manual_seed(0)
    model = nn.Sequential(*[P.PGCN(eng, f, f) for _ in range(L)]).to(dev)
    P.initiliaze_parameters(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    H = torch.rand(part.n_local, f, device=dev, generator=gen).requires_grad_(True)
    labels = part.owned.to(dev) % f

    def step():
        logits = model(H)
        loss = P.local_loss(logits, labels, n)
        opt.zero_grad()
        loss.backward()
        P.average_gradients(model)
        opt.step()
        return loss

    timer = KernelTimer(K, dev)
    for _ in range(args.warmup):
        loss = step()
        stage("warm-up step done")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.on = not args.no_kernel_timing
    # N > 1: what the boundary exchange costs and how much of it the compute stream sees (engine.ExchangeProbe: HIP events on the comm
    # stream around every round, on the compute stream around every wait for one, and around the fused gradient all-reduce)
    xprobe = None
    paced = bool(args.emulate_rank and part.size > 1 and args.pace_exchange > 0)
    if (world > 1 or paced) and not args.no_kernel_timing and (paced or not (args.emulate_rank and part.size > 1)):
        xprobe = eng.probe = engine.ExchangeProbe(dev)
        xprobe.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.on = False
    exchange_report = None
    if xprobe is not None:
        xprobe.on = False
        exchange_report = exchange_summary(xprobe, P, dev, world)
        eng.probe = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        P._all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(loss)

    ms_per_step = 1e3 * elapsed / args.steps
    emul = bool(args.emulate_rank and part.size > 1)
    nnz_job = part.nnz_local if emul else nnz           # an emulated rank aggregates only ITS entries
    edges_per_s = 2 * L * nnz_job * args.steps / elapsed

    # ---- roofline of the dominant kernel (local-block forward SpMM) -----------------
    roofline = None
    avg_ms, launches = timer.summary(id(eng.A_loc), f)
    if avg_ms:
        alg = eng.A_loc.alg_bytes(f)
        achieved = alg / (avg_ms * 1e-3)
        traffic, traffic_note = pmc_traffic_for(args, world, f)
        fpass = K.fpass
        if fpass == "auto":
            fpass = "64" if (f > 64 and eng.A_loc.ncols * f * 4 >= (96 << 20) and eng.A_loc.col.numel() >= 8_000_000) else "0"
        kname = "A_loc.H forward SpMM = spmm_tasks_kernel<%s,4,1,1> (gather part%s)" % (
            {"64": "16"}.get(fpass, "32") if f > 64 else "16", {"64": ", 64 features per pass"}.get(fpass, "") if f > 64 else "")
        if getattr(eng.A_loc, "strip", None) is not None:
            kname += " + spmm_strip_kernel (512x128 strip tiles, async LDS pipeline, %.0f%% of the entries)" % (
                100.0 * eng.A_loc.strip.nnz / max(eng.A_loc.nnz, 1))
        if eng.A_loc.core is not None:
            kname += " + spmm_core_kernel<4> (LDS-tiled dense core, %.0f%% of the entries)" % (
                100.0 * eng.A_loc.core.nnz / max(eng.A_loc.nnz, 1))
        if getattr(eng.A_loc, "dense3", None) is not None:
            kname += " + spmm_split_panels_kernel + spmm_dense3_kernel<4> (512x128 blocks on the bf16 matrix cores, three-plane split at fp32 accuracy, %.0f%% of the entries)" % (
                100.0 * eng.A_loc.dense3.nnz / max(eng.A_loc.nnz, 1))
        fx = "fix-up"
        if partition._T.lanes and eng.A_loc.nnz >= partition._T.lanes_min_nnz and "/" in partition._T.lanes:
            kname += " + %s; one launch group on two streams (lanes %s), timed as a whole" % (fx, partition._T.lanes)
        else:
            kname += " + %s; one launch group, timed as a whole" % fx
        roofline = {"bound": "hbm", "kernel": kname,
                    "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_note": traffic_note,
                    "alg_bytes_per_launch": alg, "avg_launch_ms": avg_ms, "launches_timed": launches,
                    "frac_of_measured_copy_ceiling_6.29TBs": achieved / 6.29e12,
                    "gather_model_GBs": 4.0 * f * eng.A_loc.nnz / (avg_ms * 1e-3) / 1e9,
                    "kernel_edges_per_s": eng.A_loc.nnz / (avg_ms * 1e-3)}
        bavg, bl = timer.summary(id(eng.A_loc_T), f)
        if bavg:
            roofline["avg_launch_ms_backward_AT"] = bavg
        # per-kernel split of the launch group: a few extra forward launches OUTSIDE the timed region
        real_lib, real_lane = K.lib, K.single_lane
        try:
            K.spmm = timer._spmm
            K.lib = SplitTimer(real_lib)
            K.single_lane = True                # (the split is of the kernels one after the other, whatever tuning.lanes says)
            eng.A_loc.launch_cache.clear()
            Csplit = torch.empty((part.n_local, f), device=dev)
            with torch.no_grad():
                for _ in range(5):
                    K.spmm(eng.A_loc, H.detach(), Csplit)
            torch.cuda.synchronize()
            roofline["split_us"] = K.lib.summary_us()         # the kernels one after the other on one stream (lanes off)
        except Exception as e:
            roofline["split_us"] = {"error": repr(e)}
        finally:                                # whatever happened: what runs after this (--graph) sees the configured library and lanes
            K.lib, K.single_lane = real_lib, real_lane
            eng.A_loc.launch_cache.clear()
    halo_groups = None
    if part.size > 1:       # the halo launch groups of this rank (A_halo[r] . slab, one per exchange round)
        halo_groups = []
        for r, Ah in enumerate(eng.A_halo):
            havg, hl = timer.summary(id(Ah), f)
            if havg:
                halg = 8 * Ah.nnz + 8 * (Ah.nrows + 1) + 4 * f * (part.round_recv_off[r][-1] - part.round_recv_off[r][0]) \
                    + 2 * 4 * f * part.n_local                   # C is read and written (accumulate)
                htraffic, _ = pmc_traffic_for(args, world, f, "halo%d" % r)
                halo_groups.append({"round": r, "nnz": Ah.nnz, "avg_launch_ms": havg, "launches_timed": hl, "traffic": htraffic,
                                    "alg_bytes_per_launch": halg, "achieved": halg / (havg * 1e-3) / 1e9, "unit": "GB/s",
                                    "frac": halg / (havg * 1e-3) / HBM_PEAK, "ps_per_entry": 1e9 * havg / max(Ah.nnz, 1)})
        if roofline is not None:
            roofline["ps_per_entry"] = 1e9 * roofline["avg_launch_ms"] / max(eng.A_loc.nnz, 1)

    out = {
        "metric": "edges aggregated/sec (%s-shaped %d-layer GCN f=%d, full training epoch%s)" % (
            args.workload.capitalize(), L, f,
            "; COMPUTE of rank %s alone on one GPU, exchange = no-op" % args.emulate_rank if emul else ""),
        "value": edges_per_s, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": info["data"],
        "config": {"workload": ("%s n=%d nnz=%d (incl. self loops), %d-layer GCN f=%d, "
                                "1D partition (%s) over %d GPU(s), 1 step = 1 epoch (fwd+loss+bwd+allreduce+Adam)")
                               % (info["source"], n, nnz, L, f, partition_name, part.size),
                   "n": n, "nnz": nnz, "f": f, "layers": L, "spmm_per_epoch": 2 * L,
                   "partition": partition_name, "generator": args.generator, "source": info["source"],
                   "emulated_rank": args.emulate_rank if emul else None,
                   "rank_shape": {"n_local": part.n_local, "n_halo": part.n_halo, "n_send": part.n_send,
                                  "nnz_local_block": part.A_loc.nnz, "nnz_halo_blocks": sum(a.nnz for a in part.A_halo)},
                   "exchange": exch.name if exch else "none",
                   "xcd_slices": eng.A_loc.nslices, "chunk": K.chunk,
                   "core_tile_fill_min": partition.CORE_TAU,
                   "bf16x3_block_fill_min": partition.DENSE3_TAU if partition.DENSE3_ON else None,
                   "exchange_rounds": part.rounds, "vertex_order": {k: v for k, v in (getattr(part, "order_info", None) or {}).items() if not k.startswith("_")},
                   "dense_gemm": ("x.W^T and g.W: the package's own matrix-core kernels (see dense_fused); dW = Gm^T.AH: PyTorch's batched "
                                  "product over 64 row slabs + their sum" if int(partition._T.dense_fused) >= 2 and f <= 128 and f % 4 == 0 else
                                  ("stock rocBLAS GEMMs launched by the solution index recorded in tunableop/gfx950.csv (x.W^T, g.W); "
                                   "dW: PyTorch's default pick" if not partition._T.gemm_tunableop else
                                   "stock rocBLAS / hipBLASLt via PyTorch, kernel per shape picked by TunableOp in set-up") if gemm_tuned
                                  else "stock rocBLAS / hipBLASLt via PyTorch (default pick)"),
                   "dense_fused": {0: "off (library GEMM + clamp / mask passes)",
                                   1: "relu(x.W^T) by gemm/pgcn_dense.hip (bf16-split MFMA, fp32 accuracy)",
                                   2: "relu(x.W^T) and (g (.) mask).W by gemm/pgcn_dense.hip (bf16-split MFMA, fp32 accuracy)"
                                   }.get(int(partition._T.dense_fused), str(partition._T.dense_fused)),
                   "strip_tiles": {"min_entries": partition.STRIP_MIN, "layer_min": partition.STRIP_LAYER_MIN,
                                   "whole_graphs_from_nnz": partition._T.strip_big_nnz, "min_entries_big": partition._T.strip_min_big,
                                   "layer_min_big": partition._T.strip_layer_min_big}
                   if partition.STRIP_ON else None,
                   "launch_lanes": ({"lanes": partition._T.lanes, "from_nnz": partition._T.lanes_min_nnz,
                                     "on_for_this_block": bool(partition._T.lanes and eng.A_loc.nnz >= partition._T.lanes_min_nnz)}
                                    if hasattr(partition._T, "lanes") else None)},
        "roofline": roofline, "ms_per_epoch": ms_per_step, "loss": loss_val, "setup_s": setup_s,
        "setup_stages_s": {"process_start_to_main": round(t_start - PROCESS_T0, 3), **{k: v for k, v in stages}},
    }
    if halo_groups is not None:
        out["halo_groups"] = halo_groups
    if world > 1:
        vol = torch.tensor([eng.stats["send_volume"]], dtype=torch.float64, device=dev)
        P._all_reduce(vol)
        out["exchange_rows_total"] = float(vol)
        out["selftest"] = selftest
    if exchange_report is not None:
        out["exchange"] = exchange_report
    out["config"]["timed_region"] = "%d eagerly launched training steps" % args.steps
    if args.graph == "on" or (args.graph == "auto" and world > 1 and dev.type == "cuda"):
        # LAST thing that touches the device: the eager line above is complete.  A replay whose collectives never finish cannot be
        # recovered from, but it must not cost the eager line either: on a deadline every rank leaves, rank 0 with its line printed.
        def bail(what):
            out["graph_replay"] = {"captured": True, "ranks": world, "eager_ms_per_step": ms_per_step,
                                   "error": "%s did not complete within its deadline; the eager line stands" % what}
            out["replay_failed"] = True            # (explicit flag: the launcher sees exit code 0 because the eager line IS complete)
            if rank == 0:
                out.setdefault("cpu_baseline", None)
                print(json.dumps(out), flush=True)
            sys.stderr.write("pgcn bench: rank %d leaves after a stuck graph replay (%s)\n" % (rank, what))
            sys.stderr.flush()
            os._exit(0)
        out["graph_replay"] = graph_replay(model, lambda: nn.Sequential(*[P.PGCN(eng, f, f) for _ in range(L)]).to(dev),
                                           H, labels, n, P, args.steps, dev, ms_per_step, world, on_timeout=bail)
        gr = out["graph_replay"]
        if gr.get("captured") and gr.get("ms_per_step"):
            # ONE method for `value` at every N (r06; the r05 line took the faster of two): the eagerly launched steps.  The same K steps as K
            # replays of one captured HIP graph of the whole step (RCCL calls included; MAX over ranks) are reported beside it -- at
            # N > 1 a rank's step is ~150 launches of ~20 us and the host-side enqueue is 11-13 % of it, which the replay removes.
            gr["value"] = 2 * L * nnz_job / (gr["ms_per_step"] * 1e-3)
            gr["unit"] = out["unit"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not emul:
        try:
            out["cpu_baseline"] = cpu_baseline(part, f, L, args.cpu_budget)
        except Exception as e:                     # the host-side comparison leg must never cost the GPU line
            out["cpu_baseline"] = {"value": None, "unit": "edges aggregated/s", "kind": "port", "error": repr(e)[:300]}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        exch.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
