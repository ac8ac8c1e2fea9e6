#!/usr/bin/env python3
"""More golden vectors FROM THE REFERENCE ITSELF (/root/reference/GPU/PGCN.py, unmodified, gloo on the CPU), produced
with the machinery of make_golden.py (same seeds, same drivers) on the Cora shape of BASELINE.json configs[0]:

  cora.A.mtx           synth.make_graph("cora") -- 2 708 vertices, 10 556 directed edges + self loops, normalised --
                       written once as a MatrixMarket DATA fixture (no data sets in this image; the portable stream
                       makes it reproducible), with seeded 1- / 2- / 4-way part vectors in the reference's format
  ref_coraA_rp2        maps + PSpMM forward AND backward, two ranks
  ref_coraA_rp4        maps + PSpMM forward, FOUR ranks (the first four-way case held against the reference's maps)
  ref_gat_coraA        two dense PGAT layers of the reference (GPU/PGAT.py:138-151), f = 8: outputs and all gradients
  ref_train_coraA      the body of run() at P = 1: 2 layers, f = 16 -- losses and final weights, cross-checked against
                       the stdout of the unmodified ref.run()

Kept apart from make_golden.py so that that script keeps regenerating its own files bit for bit.
Build container only.  Usage:  python tests/golden/make_golden_more.py"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch.multiprocessing as mp
from scipy.io import mmwrite

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402
from conftest import pkg  # noqa: E402


def main():
    synth, io_ = pkg("synth"), pkg("pargcn_io")
    n, row, col, val = synth.make_graph("cora", seed=0)
    mtx = os.path.join(HERE, "cora.A.mtx")
    mmwrite(mtx, sp.coo_matrix((val.numpy(), (row.numpy(), col.numpy())), shape=(n, n)), precision=9,
            comment="synth.make_graph('cora', seed=0): the Cora shape of BASELINE configs[0], synthetic stand-in")
    rng = np.random.default_rng(mg.SEED + 1)
    for k in (1, 2, 4):
        io_.write_partvec(os.path.join(HERE, "cora.A.mtx.%d.rp" % k), rng.integers(0, k, n) if k > 1 else np.zeros(n, np.int64))
    port = 29760
    for name, k in (("ref_coraA_rp2", 2), ("ref_coraA_rp4", 4)):
        port += 1
        mg.run_case(name, mtx, os.path.join(HERE, "cora.A.mtx.%d.rp" % k), k, port)
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=mg.train_case, args=("ref_train_coraA", mtx, os.path.join(HERE, "cora.A.mtx.1.rp"), 2, 16, port + 1))
    p.start()
    p.join()
    assert p.exitcode == 0
    # GAT (row N3): outputs and ALL gradients of two of the reference's own dense PGAT layers (GPU/PGAT.py) on the same graph
    import torch
    import make_golden_gat as mgg
    torch.set_num_threads(1)
    mgg.operator_level(mgg.load_ref(), "ref_gat_coraA", "cora.A.mtx", 8, 2)


if __name__ == "__main__":
    main()
