"""Process bodies for the multi-process (gloo, CPU) tests.  Spawned by test_engine_gloo.py."""
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, P, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(P))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=P)


def _pgcn_module(rank, P):
    from conftest import pkg
    from oracle_kernels import OracleKernels
    M = pkg("PGCN")
    M._kernel_provider = OracleKernels()       # test-only checker-backed kernels
    M.myrank, M.world_size, M.device = rank, P, torch.device("cpu")
    M._exchanger = None
    return M


def pspmm_worker(rank, P, port, path_A, path_pv, f, seed, q):
    """PSpMM.apply forward + backward on the golden inputs, owned rows only."""
    from conftest import golden_inputs, read_partvec
    from scipy.io import mmread
    _init(rank, P, port)
    M = _pgcn_module(rank, P)
    A = mmread(path_A)
    part = read_partvec(path_pv)
    M.send_map, M.recv_map = M.compute_communication_maps(A, part, rank, P)
    eng = M.get_partitiont_of_adjacency_matrix(A, part, rank)
    M.init_stats()
    own = eng.part.owned.numpy()
    Hfull, Gfull = golden_inputs(A.shape[0], f, seed)
    H = torch.tensor(Hfull[own], requires_grad=True)
    from conftest import pkg as _pkg
    probe = eng.probe = _pkg("engine").ExchangeProbe(torch.device("cpu"))      # what bench.py attaches at N > 1 (r06)
    probe.on = P > 1
    out = M.PSpMM.apply(eng, H)
    M._sync_stats(eng)                       # counters live in the engine; run() publishes them
    stats_fwd = {k: int(v) for k, v in M.stats.items()}
    out.backward(torch.tensor(Gfull[own]))
    if P > 1:
        g = torch.ones(3 * f)
        eng.allreduce_sum(g)
        assert torch.equal(g, torch.full((3 * f,), float(P)))
    probe.on = False
    exchange = probe.summary()
    eng.probe = None
    # the standalone communicate_fgm entry point
    halo = M.communicate_fgm(H.detach(), backward=False)
    ok_halo = bool(np.array_equal(halo.numpy(), Hfull[eng.part.halo_global.numpy()]))
    q.put({"rank": rank, "own": own, "fwd": out.detach().numpy(), "bwd": H.grad.numpy(),
           "stats_fwd": stats_fwd, "stats_all": {k: int(v) for k, v in M.stats.items()},
           "ok_halo": ok_halo, "exchange": exchange, "rounds": eng.rounds,
           "round_send_off": eng.round_send_off, "round_recv_off": eng.round_recv_off,
           "send_map": {k: v.numpy() for k, v in M.send_map.items()}})
    dist.barrier()
    dist.destroy_process_group()


def run_worker(rank, P, port, path_A, path_pv, nlayers, f, seed, q):
    """The drop-in's run() end to end (PGCN.py:162-238) over gloo."""
    _init(rank, P, port)
    M = _pgcn_module(rank, P)
    torch.manual_seed(seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model = M.run(rank, P, nlayers, f, path_A, path_pv, "gloo")
    q.put({"rank": rank, "stdout": buf.getvalue(),
           "weights": [m.linear.weight.detach().numpy() for m in model]})
    dist.barrier()
    dist.destroy_process_group()


def pargcn_worker(rank, P, port, path_A, path_pv, d, seed, q):
    """Parallel-GCN semantics (pargcn.train) over gloo with the engine."""
    from conftest import pkg, read_partvec
    from oracle import oracle
    from scipy.io import mmread
    import scipy.sparse as sp
    _init(rank, P, port)
    M = _pgcn_module(rank, P)
    A = oracle.normalize_adjacency(mmread(path_A))
    A = ((A + A.T) * 0.5).tocoo().astype(np.float32)
    part = read_partvec(path_pv)
    eng = M.get_partitiont_of_adjacency_matrix(A, part, rank)
    n = A.shape[0]
    rng = np.random.default_rng(seed)
    L = len(d) - 1
    W = {l: torch.tensor((rng.random((d[l], d[l + 1]), dtype=np.float32) * 2 - 1) *
                         np.float32(np.sqrt(6.0 / (d[l] + d[l + 1])))) for l in range(1, L)}
    own = eng.part.owned.numpy()
    H0 = torch.ones((own.size, d[1]))
    Y = torch.zeros((own.size, d[L])); Y[:, 1] = 1
    Ym = torch.zeros((own.size, d[L]), dtype=torch.uint8); Ym[:, 1] = 1
    ar = (lambda t: dist.all_reduce(t)) if P > 1 else None
    errs, Wn, Hl = pkg("pargcn").train(eng, d, W, H0, Y, Ym, epochs=3, alpha=0.01, allreduce=ar)
    q.put({"rank": rank, "own": own, "errs": errs, "W": {l: w.numpy() for l, w in Wn.items()},
           "Hl": Hl.numpy(), "stats": dict(eng.stats)})
    dist.barrier()
    dist.destroy_process_group()


def run_worker_gpu(rank, P, port, path_A, path_pv, nlayers, f, seed, q):
    """run() with the REAL HIP kernels: P processes share the one GPU, gloo as transport
    (RCCL refuses two ranks on one device; everything else is the production path)."""
    _init(rank, P, port)
    from conftest import pkg
    M = pkg("PGCN")
    M._kernel_provider = None
    M._exchanger = None
    torch.manual_seed(seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model = M.run(rank, P, nlayers, f, path_A, path_pv, "gloo")
    q.put({"rank": rank, "stdout": buf.getvalue(), "provider": type(M._kernel_provider).__name__,
           "overlap": bool(M._engine_current.overlap),
           "weights": [m.linear.weight.detach().cpu().numpy() for m in model]})
    dist.barrier()
    dist.destroy_process_group()


def minibatch_worker(rank, P, port, path_A, path_pv, f, bs, seed, gpu, q):
    """PGCN_minibatch.run(); gpu=True uses the real kernels (processes share cuda:0), else the checker."""
    _init(rank, P, port)
    from conftest import pkg
    M = pkg("PGCN")
    MB = pkg("PGCN_minibatch")
    if gpu:
        M._kernel_provider = None
    else:
        from oracle_kernels import OracleKernels
        M._kernel_provider = OracleKernels()
    M._exchanger = None
    torch.manual_seed(seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model = MB.run(rank, P, 3, f, path_A, path_pv, "gloo", bs)
    q.put({"rank": rank, "stdout": buf.getvalue(),
           "weights": [m.linear.weight.detach().cpu().numpy() for m in (model.gcn1, model.gcn2, model.gcn3)]})
    dist.barrier()
    dist.destroy_process_group()


# ---- GAT path ---------------------------------------------------------------------------------
def _pgat_module(rank, P, mode, heads, gpu=False):
    from conftest import pkg
    M = pkg("PGAT")
    if gpu:                                    # the real HIP kernels; processes share cuda:0
        M._kernel_provider = None
        M.myrank, M.world_size, M.device = rank, P, torch.device("cuda:0")
        torch.cuda.set_device(0)
    else:
        from oracle_kernels import OracleKernels
        M._kernel_provider = OracleKernels()   # test-only checker-backed kernels
        M.myrank, M.world_size, M.device = rank, P, torch.device("cpu")
    M.mode, M.heads = mode, heads
    M._exchanger = None
    return M


def gat_layers_worker(rank, P, port, path_A, path_pv, mode, heads, f, L, seed, q, gpu=False):
    """L PGAT layers + run()'s objective on seeded inputs / parameters: outputs and every gradient,
    owned rows only."""
    from conftest import read_partvec
    from scipy.io import mmread
    _init(rank, P, port)
    M = _pgat_module(rank, P, mode, heads, gpu)
    dev = M.device
    A = mmread(path_A)
    n = A.shape[0]
    part = read_partvec(path_pv)
    M.send_map, M.recv_map = M.compute_communication_maps(A, part, rank, P)
    eng = M.get_partitiont_of_adjacency_matrix(A, part, rank)
    own = eng.part.owned.numpy()
    rng = np.random.default_rng(seed)
    Hfull = (rng.random((n, f), dtype=np.float32) * 2 - 1)
    H = torch.tensor(Hfull[own], requires_grad=True, device=dev)
    layers = [M.PGAT(eng, f, f).to(dev) for _ in range(L)]
    with torch.no_grad():
        for layer in layers:
            layer.linear.weight.copy_(torch.from_numpy((rng.standard_normal((f, f)) * 0.4).astype(np.float32)))
            layer.attention.copy_(torch.from_numpy((rng.standard_normal(tuple(layer.attention.shape)) * 0.4).astype(np.float32)))
    x, outs = H, []
    for layer in layers:
        x = layer(x)
        outs.append(x)
    labels = torch.from_numpy(own).to(dev) % f
    loss = M.local_loss(x, labels, n)
    loss.backward()
    # the standalone Comm entry point: halo rows of H, and its backward (accumulating unpack)
    H2 = torch.tensor(Hfull[own], requires_grad=True, device=dev)
    halo = M.Comm.apply(H2)
    ok_halo = bool(np.array_equal(halo.detach().cpu().numpy(), Hfull[eng.part.halo_global.numpy()]))
    halo.sum().backward()
    q.put({"rank": rank, "own": own, "outs": [o.detach().cpu().numpy() for o in outs], "loss": float(loss.detach()),
           "dH": H.grad.cpu().numpy(), "dW": [l.linear.weight.grad.cpu().numpy() for l in layers],
           "da": [l.attention.grad.cpu().numpy() for l in layers], "ok_halo": ok_halo,
           "comm_grad": H2.grad.cpu().numpy(), "n_send_rows": int(eng.n_send),
           "fused": [bool(l._state.fused) for l in layers],
           "provider": type(M._kernel_provider).__name__})
    dist.barrier()
    dist.destroy_process_group()


def gat_run_worker(rank, P, port, path_A, path_pv, mode, heads, nlayers, f, seed, epochs, q, gpu=False):
    """The drop-in's run() end to end (PGAT.py:165-233) over gloo."""
    _init(rank, P, port)
    M = _pgat_module(rank, P, mode, heads, gpu)
    torch.manual_seed(seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model = M.run(rank, P, nlayers, f, path_A, path_pv, "gloo", epochs=epochs)
    q.put({"rank": rank, "stdout": buf.getvalue(), "provider": type(M._kernel_provider).__name__,
           "params": {k: v.detach().cpu().numpy() for k, v in model.named_parameters()}})
    dist.barrier()
    dist.destroy_process_group()


def partition_local_worker(rank, P, port, path_A, path_pv, rounds, q):
    """build_partition_local from this rank's own rows (native loader) vs build_partition from the global COO."""
    from conftest import pkg, read_partvec
    from scipy.io import mmread
    _init(rank, P, port)
    partition, ingest = pkg("partition"), pkg("ingest")
    part = read_partvec(path_pv)
    mine = ingest.load_partition(path_A, part, rank)                      # only the rows this rank owns
    loc = partition.build_partition_local(torch.from_numpy(mine.row), torch.from_numpy(mine.col), torch.from_numpy(mine.data),
                                          mine.shape[0], torch.tensor(part), rank, P, rounds=rounds)
    A = mmread(path_A).tocoo()
    glo = partition.build_partition(torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64)),
                                    torch.from_numpy(A.data.astype(np.float32)), A.shape[0], torch.tensor(part), rank, P,
                                    rounds=rounds)

    def same_csr(a, b):
        if a is None or b is None:
            return a is b
        ra, ca, va = a.to_coo()
        rb, cb, vb = b.to_coo()
        ok = a.nrows == b.nrows and a.ncols == b.ncols and torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col, b.col)
        ok = ok and torch.equal(a.val, b.val) and torch.equal(ra, rb) and torch.equal(ca, cb) and torch.equal(va, vb)
        ok = ok and (a.row_map is None) == (b.row_map is None) and (a.row_map is None or torch.equal(a.row_map, b.row_map))
        return bool(ok)

    checks = {
        "scalars": (loc.n, loc.rank, loc.size, loc.nnz_global) == (glo.n, glo.rank, glo.size, glo.nnz_global),
        "owned": torch.equal(loc.owned, glo.owned), "send_idx": torch.equal(loc.send_idx, glo.send_idx),
        "send_owner": torch.equal(loc.send_owner, glo.send_owner), "halo_owner": torch.equal(loc.halo_owner, glo.halo_owner),
        "halo_global": torch.equal(loc.halo_global, glo.halo_global), "send_global": torch.equal(loc.send_global, glo.send_global),
        "offs": loc.round_send_off == glo.round_send_off and loc.round_recv_off == glo.round_recv_off,
        "A_loc": same_csr(loc.A_loc, glo.A_loc), "A_loc_T": same_csr(loc.A_loc_T, glo.A_loc_T),
        "A_halo": len(loc.A_halo) == len(glo.A_halo) and all(same_csr(a, b) for a, b in zip(loc.A_halo, glo.A_halo)),
        "A_halo_T": all(same_csr(a, b) for a, b in zip(loc.A_halo_T, glo.A_halo_T)),
        "unpack": all(same_csr(a, b) for a, b in zip(loc.unpack, glo.unpack)),
    }
    q.put({"rank": rank, "checks": {k: bool(v) for k, v in checks.items()}, "nnz_local": int(mine.nnz)})
    dist.barrier()
    dist.destroy_process_group()


def bench_inputs_worker(rank, P, port, kind, path, path_pv, q):
    """bench.acquire_partition on one of the input kinds of bench.py (shards / mtx) under gloo, device = cpu."""
    import argparse
    _init(rank, P, port)
    import bench
    args = argparse.Namespace(shards=path if kind == "shards" else None, mtx=path if kind == "mtx" else None,
                              partvec=path_pv, emulate_rank=None, workload="none", generator="rmat", real=False)
    part, info = bench.acquire_partition(args, rank, P, torch.device("cpu"), lambda m: None)
    r, c, v = part.A_loc.to_coo()
    key = torch.argsort(r * part.n_local + c, stable=True)
    q.put({"rank": rank, "owned": part.owned.numpy(), "halo_global": part.halo_global.numpy(),
           "send_global": part.send_global.numpy(), "nnz": info["nnz"], "n": info["n"], "partition": info["partition"],
           "loc": (r[key].numpy(), c[key].numpy(), v[key].numpy()),
           "nnz_halo": sum(a.nnz for a in part.A_halo)})
    dist.barrier()
    dist.destroy_process_group()


def pargcn_main_worker(rank, P, port, directory, seed, q, provider="oracle"):
    """pargcn.main (the `grbgcn -p DIR -c CONFIG` command line of Parallel-GCN/main.c:120-165) with world > 1
    over gloo: checker-backed kernels ("oracle", "plans"), or -- provider "hip" -- the product's own choice, i.e. the
    REAL HIP kernels on every rank, the P processes sharing the one GPU (gloo transport, host-staged)."""
    _init(rank, P, port)
    from conftest import pkg
    from oracle_kernels import OracleKernels, PlanKernels
    os.environ["PGCN_SEED"] = str(seed)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(P), "0"
    buf = io.StringIO()
    kernels = None if provider == "hip" else (PlanKernels() if provider == "plans" else OracleKernels())
    errs, Wn, Hout, part = pkg("pargcn").main(["-p", directory, "-c", os.path.join(directory, "config")], kernels=kernels, out=buf)
    q.put({"rank": rank, "stdout": buf.getvalue(), "errs": [float(e) for e in errs],
           "W": {l: w.cpu().numpy() for l, w in Wn.items()}, "own": part.owned.numpy(), "H": Hout.cpu().numpy(),
           "n_send": part.n_send, "n_halo": part.n_halo,
           "targets": int(torch.unique(part.send_owner).numel()), "sources": int(torch.unique(part.halo_owner).numel())})
    dist.barrier()
    dist.destroy_process_group()


def bench_selftest_worker(rank, P, port, break_it, q):
    """bench.multirank_selftest over gloo with the checker-backed kernels: the first-contact check of an N-rank job
    (P ranks through the real engine + exchange against the same kernels on one rank).  ``break_it``: an exchanger that
    delivers a wrong row -- the check must notice."""
    _init(rank, P, port)
    import importlib
    from conftest import pkg, ROOT
    from oracle_kernels import OracleKernels
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    engine = pkg("engine")
    ex = engine.make_exchanger(rank, P, torch.device("cpu"))
    if break_it:
        real = ex.alltoallv

        def bad(send, send_off, recv, recv_off, f):
            real(send, send_off, recv, recv_off, f)
            if rank == 1 and recv.shape[0] > 2:
                recv[2] += 1.0                      # one received row is off
        ex.alltoallv = bad
    try:
        rec = bench.multirank_selftest(rank, P, torch.device("cpu"), OracleKernels(), ex, n=3000, nnz=60000, f=8)
        q.put({"rank": rank, "ok": True, "rec": rec})
    except RuntimeError as e:
        q.put({"rank": rank, "ok": False, "msg": str(e)[:300]})
    dist.barrier()
    dist.destroy_process_group()
