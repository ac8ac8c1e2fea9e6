"""Child process of tests/test_plan_cpu.py::test_non_default_tunings_keep_the_plans_exact: PGCN_TUNING is read ONCE at
import, so every tuning runs in its own interpreter.  Builds rank `rk` of `P` of a dense-cornered graph under the tuning
in the environment, executes its launch plans on the CPU and prints which parts exist and the forward error."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import pkg  # noqa: E402
from plan_interpreter import HostPlanner, run_plan  # noqa: E402


def main():
    partition, synth = pkg("partition"), pkg("synth")
    n, row, col, val = synth.make_graph(9000, 2400000, seed=1)
    A = sp.csr_matrix((val.numpy().astype(np.float64), (row.numpy(), col.numpy())), shape=(n, n))
    H = np.random.default_rng(0).standard_normal((n, 2))
    AH = A @ H
    K = HostPlanner()
    for P, rk in ((1, 0), (3, 2)):
        pv = torch.zeros(n, dtype=torch.int64) if P == 1 else synth.random_partvec(n, P, seed=0)
        pt = partition.build_partition(row, col, val, n, pv, rk, P)
        own, hg = pt.owned.numpy(), pt.halo_global.numpy()
        C, _ = run_plan(K.prepare(pt.A_loc), H[own])
        C = np.nan_to_num(C)
        for a in pt.A_halo:
            C, _ = run_plan(K.prepare(a), H[hg], C0=C, accumulate=True)
        parts = "".join(ch for ch, blk in (("s", "strip"), ("c", "core"), ("3", "dense3"))
                        if any(getattr(b, blk) is not None for b in [pt.A_loc] + list(pt.A_halo)))
        print("P=%d parts=%s rounds=%d err=%.3e" % (P, parts or "-", pt.rounds, np.abs(C - AH[own]).max()))


if __name__ == "__main__":
    main()
