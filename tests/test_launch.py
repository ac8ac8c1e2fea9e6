"""Command lines end to end: `python bench.py --gpus N` (the driver's form, self-launching for N > 1) and the
drop-in `python PGCN.py -a .. -p .. -b .. -s .. -l .. -f ..` (reference: GPU/PGCN.py:256-283 main / init_process,
stdout lines :224,230,237,238,249)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, free_port, gpath

_STRIP = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "SLURM_PROCID", "SLURM_NPROCS", "MASTER_PORT",
          "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in _STRIP}
    env["MASTER_ADDR"] = "127.0.0.1"
    env.update({k: str(v) for k, v in kw.items()})
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "expected ONE JSON line, got %d:\n%s" % (len(lines), stdout[-2000:])
    return json.loads(lines[0])


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the launch mechanics on a box without a GPU")
def test_bench_self_launch_starts_every_rank_and_fails_loudly_without_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start 2 ranks itself; without a
    GPU every rank refuses to run (no CPU fallback) and says which rank it is."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cora",
                          "--steps", "1", "--warmup", "0"], env=_env(PGCN_BENCH_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "[rank 0 of 2]" in out.stderr and "[rank 1 of 2]" in out.stderr, out.stderr[-3000:]
    assert "needs an MI355X" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_smoke_entry_point():
    """__graft_entry__.smoke() -- what the driver runs before the bench -- in its own process (it changes module-level
    thresholds so that its small graph keeps every kind of tile): every SpMM kernel on the path, checked against the oracle."""
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=_env(),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stderr[-2000:] + out.stdout[-500:]


@pytest.mark.gpu
def test_bench_single_gpu_line_small_workload():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "cora",
                          "--steps", "3", "--warmup", "1", "--cpu-budget", "1"], env=_env(),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["value"] > 0
    assert rec["roofline"] and rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1
    assert rec["cpu_baseline"] and rec["cpu_baseline"]["value"] > 0
    assert rec["config"]["timed_region"].startswith("3 eagerly launched") and "graph_replay" not in rec    # (N = 1: --graph auto is off)
    assert np.isfinite(rec["loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 4])
def test_bench_self_launch_n_ranks_share_the_gpu(N):
    """The exact command form the driver uses for N = 1, with N > 1 and nothing else: the ranks are started by
    bench.py itself (here they share the one GPU and talk over gloo; real kernels, comm-stream overlap on)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--workload", "cora",
                          "--steps", "3", "--warmup", "1"], env=_env(PGCN_BENCH_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    assert rec["n_gpus"] == N and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["config"]["partition"] != "none" and rec["exchange_rows_total"] > 0
    # N > 1: the graph replay is attempted automatically (r05); over gloo the exchange stages through the host and cannot be
    # captured, so every rank agrees not to replay and the eager steps stay the line's value
    assert "graph_replay" in rec and rec["graph_replay"].get("captured") is False
    # r06: `value` is ALWAYS the eagerly launched steps (one method at every N); a replay is reported beside it
    assert rec["config"]["timed_region"].startswith("3 eagerly launched") and "eager" not in rec and "replay_failed" not in rec
    assert np.isfinite(rec["loss"])
    # r06: the line explains its exchange -- per direction and round: bytes, the largest peer segment, ms, exposed ms, link rate
    ex = rec["exchange"]
    L, f = rec["config"]["layers"], rec["config"]["f"]
    for tag in ("forward", "backward"):
        assert [e["round"] for e in ex[tag]] == list(range(len(ex[tag]))) and len(ex[tag]) >= 1
        for e in ex[tag]:
            assert e["calls"] == 3 * L and e["bytes_out"] >= 0 and e["bytes_in"] >= 0 and e["max_peer_bytes"] <= max(e["bytes_out"], e["bytes_in"])
            assert e["ms"] >= 0 and 0 <= e["exposed_ms"] and e["ms_max_over_ranks"] >= e["ms"] - 1e-9
            assert abs(e["frac_of_153GBs"] - e["GBs_per_link"] / 153.0) < 1e-9
        assert sum(e["bytes_out"] for e in ex[tag]) > 0
    assert ex["allreduce"]["calls"] == 3 and ex["allreduce"]["bytes"] == L * f * f * 4


def _run_pgcn_cli(ranks, size, backend, mtx, pv, L, f, port):
    procs = []
    for r in ranks:
        env = _env(SLURM_NPROCS=size, SLURM_PROCID=r, MASTER_PORT=port, WORLD_SIZE=size)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "PGCN.py"), "-a", gpath(mtx), "-p", gpath(pv),
                                       "-b", backend, "-s", str(size), "-l", str(L), "-f", str(f)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(o)
    return outs


def _check_stdout(out, rank, size, L, n_epochs=4):
    """The reference's lines: :249 process-group echo, :224 per-epoch loss (rank 0), :230 stats dict (every rank),
    :237 elapsed and :238 totals (rank 0)."""
    assert re.search(r"\[\d+\] Initializing process group with: \{'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '\d+', "
                     r"'RANK': '%d', 'WORLD_SIZE': '%d'\}" % (rank, size), out), out
    m = re.search(r"\{'send_volume': tensor\((\d+)\), 'recv_volume': tensor\((\d+)\), 'send_nmsg': tensor\((\d+)\), "
                  r"'recv_nmsg': tensor\((\d+)\)\}", out)
    assert m, out
    losses = [float(x) for x in re.findall(r"Epoch (?:0000\d) \| Loss ([0-9.]+)", out)]
    if rank == 0:
        assert len(losses) == n_epochs and all(np.isfinite(losses))
        assert re.search(r"Elapsed time \d+\.\d{4}", out)
        t = re.search(r"total_vol: (\d+) total_nmsg: (\d+)", out)
        assert t, out
        return [int(x) for x in m.groups()], [int(x) for x in t.groups()], losses
    assert not losses and "Elapsed time" not in out and "total_vol" not in out
    return [int(x) for x in m.groups()], None, losses


@pytest.mark.gpu
def test_pgcn_main_single_rank_rccl_backend():
    """a10: `python PGCN.py ... -b nccl -s 1` with SLURM_* in the environment, through main() -> spawn ->
    init_process(nccl = RCCL) -> run()."""
    L, f = 3, 16
    (out,) = _run_pgcn_cli([0], 1, "nccl", "karate.A.mtx", "karate.mtx.1.rp", L, f, free_port())
    stats, tot, losses = _check_stdout(out, 0, 1, L)
    assert stats == [0, 0, 0, 0] and tot == [0, 0]
    assert losses[-1] <= losses[0]


@pytest.mark.gpu
def test_pgcn_main_three_ranks_share_the_gpu():
    """a10 with P = 3 (one invocation per rank, as under srun): KATs of the message statistics
    (5 epochs x 2L exchanges x rows / peers, GPU/PGCN.py:105-114)."""
    from scipy.io import mmread
    import scipy.sparse as sp
    from oracle import oracle
    from conftest import read_partvec
    L, f, P = 2, 16, 3
    outs = _run_pgcn_cli(range(P), P, "gloo", "gemat11p.A.mtx", "gemat11.mtx.3.hp", L, f, free_port())
    A = sp.csr_matrix(mmread(gpath("gemat11p.A.mtx"))).astype(np.float32)
    part = read_partvec(gpath("gemat11.mtx.3.hp"))
    rows_total = 0
    for r in range(P):
        stats, tot, _ = _check_stdout(outs[r], r, P, L)
        smap, rmap = oracle.communication_maps(A, part, r, P)
        rows_out, rows_in = sum(v.size for v in smap.values()), sum(v.size for v in rmap.values())
        nx = 5 * L * 2
        # forward exchanges send my boundary rows and receive the halo rows, the reverse exchanges of backward
        # send one partial row per halo row and receive one per boundary row (GPU/PGCN.py:91-97 swaps the maps)
        half = nx // 2
        assert stats == [(rows_out + rows_in) * half, (rows_in + rows_out) * half, (P - 1) * nx, (P - 1) * nx]
        rows_total += rows_out
        if r == 0:
            totals = tot
    assert totals == [rows_total * 5 * L * 2, P * (P - 1) * 5 * L * 2]


@pytest.mark.gpu
def test_bench_with_reference_hypergraph_part_vector():
    """`bench.py --partvec FILE`: a part vector written by the reference's GPU/hypergraph tool (PaToH) for the
    `mid` workload (tests/golden/partvec, tools/make_partvecs.py).  The exchanged volume is the tool's cut."""
    from conftest import GOLDEN
    pv = os.path.join(GOLDEN, "partvec", "mid.A.mtx.2.hp.gz")
    with open(os.path.join(GOLDEN, "partvec", "mid.stats.json")) as fh:
        stats = json.load(fh)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "mid", "--partvec", pv,
                          "--steps", "2", "--warmup", "1"], env=_env(PGCN_BENCH_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    assert rec["config"]["partition"] == "file:mid.A.mtx.2.hp.gz" and rec["n_gpus"] == 2
    rows = stats["parts"]["2"]["hp"]["boundary_rows_per_aggregation"]
    L = rec["config"]["layers"]
    assert rec["exchange_rows_total"] == rows * 2 * L * 3          # (warm-up + steps) epochs x 2L aggregations
    assert rows < stats["parts"]["2"]["rp"]["boundary_rows_per_aggregation"]


@pytest.mark.gpu
def test_bench_sbm_generator_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "mid", "--generator", "sbm",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=_env(),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    assert rec["config"]["generator"] == "sbm" and "planted-partition" in rec["config"]["workload"]
    assert rec["config"]["vertex_order"]["order"] == "community"
    assert rec["roofline"]["split_us"] and "error" not in rec["roofline"]["split_us"]


@pytest.mark.gpu
def test_bench_from_binary_shards_papers_shape_two_ranks(tmp_path):
    """BASELINE config 4 (papers100M, GCN-GP part vector, f = 64, L = 2) rehearsed at 1/64 scale: rank-local generator
    -> binary CSR shards -> `bench.py --shards PREFIX` with two ranks over gloo.  No process ever holds the global
    matrix (the reference parses all of it on every rank, GPU/PGCN.py:171); the part vector is the one the reference's
    own METIS front-end (GPU/graph/main.cpp) wrote for the union of the shards (tools/make_partvecs.py
    --generator shardstream), so the exchanged volume is that tool's cut."""
    from conftest import GOLDEN
    pv = os.path.join(GOLDEN, "partvec", "papers64.A.mtx.2.gp.gz")
    with open(os.path.join(GOLDEN, "partvec", "papers64.stats.json")) as fh:
        stats = json.load(fh)
    prefix = str(tmp_path / "papers64")
    mk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_shards.py"), "--workload", "papers", "--scale", "0.015625",
                         "--ranks", "2", "--partvec", pv, "--out", prefix, "--device", "cuda"], env=_env(),
                        capture_output=True, text=True, timeout=1200)
    assert mk.returncode == 0, mk.stderr[-3000:]
    assert os.path.exists(prefix + ".0.pgcsr") and os.path.exists(prefix + ".1.pgcsr")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shards", prefix, "--partvec", pv,
                          "--features", "64", "--layers", "2", "--steps", "2", "--warmup", "1"],
                         env=_env(PGCN_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and cfg["n"] == stats["n"] == 1735311 and cfg["nnz"] == stats["nnz"]
    assert cfg["f"] == 64 and cfg["layers"] == 2 and "binary CSR shards" in cfg["source"]
    assert cfg["partition"] == "file:papers64.A.mtx.2.gp.gz"
    rows = stats["parts"]["2"]["gp"]["boundary_rows_per_aggregation"]
    assert rec["exchange_rows_total"] == rows * 2 * 2 * 3           # (warm-up + steps) epochs x 2L aggregations
    assert rows < stats["parts"]["2"]["rp"]["boundary_rows_per_aggregation"]
    assert np.isfinite(rec["loss"]) and rec["value"] > 0 and rec["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_gat_two_ranks_share_the_gpu():
    """BASELINE config 5 is a multi-GPU GAT: `bench.py --workload reddit-gat --gpus 2` (ranks share the GPU over gloo,
    real attention kernels, Reddit-sized shards) prints the same kind of line as the GCN bench."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "reddit-gat",
                          "--steps", "2", "--warmup", "1"], env=_env(PGCN_BENCH_BACKEND="gloo"),
                         capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    assert rec["n_gpus"] == 2 and rec["config"]["heads"] == 4 and rec["config"]["head_dim"] == 64
    assert rec["config"]["partition"].startswith("random") and rec["exchange_rows_total"] > 0
    assert np.isfinite(rec["loss"]) and rec["value"] > 0
    assert rec["roofline"] and rec["roofline"]["frac"] > 0
    # r06: the dense blocks of the attention pattern run on the split structures of a shard too, and the line says so
    blocks = rec["config"]["blocks"]
    assert blocks and blocks["entries_on_blocks"] > 0.2 and {"fwd_local", "bwd_local"} <= set(blocks["structures"])
    split = rec["roofline"]["pass_split_ms"]
    assert split["backward_blocks"] > 0 and split["forward_blocks"] > 0 and split["backward_gather"] > 0


@pytest.mark.gpu
def test_bench_emulated_rank_line():
    """`--emulate-rank r/P`: one GPU runs the compute of rank r of a P-rank job (the shard shapes 2/4/8 GPUs run),
    with a roofline object for the local block and one per halo launch group."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mid", "--emulate-rank", "1/4",
                          "--steps", "3", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    assert rec["n_gpus"] == 1 and rec["config"]["emulated_rank"] == "1/4" and rec["cpu_baseline"] is None
    shape = rec["config"]["rank_shape"]
    assert shape["n_halo"] > 0 and shape["nnz_halo_blocks"] > 0 and abs(shape["n_local"] - 131072 / 4) < 2000
    assert len(rec["halo_groups"]) == rec["config"]["exchange_rounds"] and all(h["frac"] > 0 for h in rec["halo_groups"])
    assert rec["roofline"]["frac"] > 0 and "COMPUTE of rank 1/4" in rec["metric"]


@pytest.mark.gpu
def test_bench_graph_replay_of_one_training_step():
    """`--graph`: one whole training step of an emulated rank (local block + two halo launch groups ordered with
    events on two streams, loss kernels, Adam) is captured in a HIP graph and replayed: the C-ABI library never
    allocates or synchronises (include/pgcn_hip.h), so nothing in the step breaks a capture, and a replay costs the
    host next to nothing (compute side)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mid", "--emulate-rank", "0/4", "--graph",
                          "--steps", "5", "--warmup", "2"], env=_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    g = rec["graph_replay"]
    assert g["captured"], g
    assert g["host_enqueue_ms_per_step"] < 0.5 and 0 < g["ms_per_step"] < 1.5 * rec["ms_per_step"]
    assert np.isfinite(g["loss"])
