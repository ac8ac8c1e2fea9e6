"""pytest configuration: marker registration + shared fixtures/helpers."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_NAME = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pkg(sub: str = ""):
    """Import the product package (its directory name is not a Python identifier)."""
    return importlib.import_module(PKG_NAME + (("." + sub) if sub else ""))


def golden(name: str):
    arrays = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    return arrays, meta


def read_partvec(path: str):
    with open(path) as f:
        return list(map(int, f.readline().split()))


def gpath(name: str) -> str:
    return os.path.join(GOLDEN, name)


SPMM_CASES = [
    # golden name, matrix, part vector, P
    ("ref_karate_hp3", "karate.mtx", "karate.mtx.3.hp", 3),
    ("ref_karate_stchp3", "karate.mtx", "karate.mtx.3.stchp", 3),
    ("ref_karate_rp2", "karate.mtx", "karate.mtx.2.rp", 2),
    ("ref_karate_p1", "karate.mtx", "karate.mtx.1.rp", 1),
    ("ref_gemat11_hp3", "gemat11.mtx", "gemat11.mtx.3.hp", 3),
    ("ref_gemat11_rp3", "gemat11.mtx", "gemat11.mtx.3.rp", 3),
    ("ref_gemat11_rp2", "gemat11.mtx", "gemat11.mtx.2.rp", 2),
    ("ref_gemat11pA_rp2", "gemat11p.A.mtx", "gemat11.mtx.2.rp", 2),
    ("ref_gemat11pA_hp3", "gemat11p.A.mtx", "gemat11.mtx.3.hp", 3),
]
TRAIN_CASES = [
    ("ref_train_karate", "karate.mtx", "karate.mtx.1.rp"),
    ("ref_train_karateA", "karate.A.mtx", "karate.mtx.1.rp"),
    ("ref_train_gemat11pA", "gemat11p.A.mtx", "gemat11.mtx.1.rp"),
]
# tests/golden/make_golden_more.py (r03): the Cora shape of BASELINE configs[0] through the reference's GPU/PGCN.py;
# held by the CPU tests (maps, PSpMM over gloo with two and FOUR ranks, the P = 1 training run)
SPMM_CASES_MORE = [
    ("ref_coraA_rp2", "cora.A.mtx", "cora.A.mtx.2.rp", 2),
    ("ref_coraA_rp4", "cora.A.mtx", "cora.A.mtx.4.rp", 4),
]
TRAIN_CASES_MORE = [("ref_train_coraA", "cora.A.mtx", "cora.A.mtx.1.rp")]


def golden_inputs(n: int, f: int, seed: int):
    """The seeded H / G of tests/golden/make_golden.py:_worker."""
    rng = np.random.default_rng(seed)
    H = rng.random((n, f), dtype=np.float32) * 2 - 1
    G = rng.random((n, f), dtype=np.float32) * 2 - 1
    return H, G


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


OBSERVED_LOG = os.path.join(ROOT, "gpurun_out", "parity_observed.jsonl")


def held_to_fixture(case: str, what: str, got, fixture, shadow64, floor: float = 1e-5, record: bool = True):
    """north_star's bar ("within 1e-5 relative fp32") where the comparison value is itself an fp32 result of the
    reference: both the HIP result and the reference-made fixture are measured against a float64 shadow of the same
    arithmetic, and the HIP result may be at most ``floor`` away from the shadow -- or twice as far as the reference's
    own fp32 result is, where that one is already beyond the floor (its running fp32 sums, Adam's division by
    sqrt(v)).  No blanket tolerance: the observed pair goes to gpurun_out/parity_observed.jsonl (DESIGN.md section 2
    tabulates it) and into the assertion message."""
    e_got, e_fix = rel_err(got, shadow64), rel_err(fixture, shadow64)
    if record:
        try:
            os.makedirs(os.path.dirname(OBSERVED_LOG), exist_ok=True)
            with open(OBSERVED_LOG, "a") as fh:
                fh.write(json.dumps({"case": case, "what": what, "err_hip": e_got, "err_fixture": e_fix,
                                     "bound": max(floor, 2 * e_fix)}) + "\n")
        except OSError:
            pass
    assert e_got <= max(floor, 2 * e_fix), \
        "%s %s: HIP result is %.3g from the float64 shadow, the reference's own fp32 result %.3g (bound %.3g)" % (
            case, what, e_got, e_fix, max(floor, 2 * e_fix))
    return e_got, e_fix


def free_port() -> int:
    """A TCP port the kernel just handed out on 127.0.0.1 (fixed port numbers collide under pytest-xdist and with
    sockets of an earlier test still in TIME_WAIT)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
