"""Mini-batch driver: drop-in for /root/reference/GPU/PGCN-Mini-batch.py.

Same command line (``-a A.mtx -p partvec.pickle -b nccl|gloo -s ngpu -l layers -f hidden -n batch``,
PGCN-Mini-batch.py:325-351), same batch construction (``random.seed(1)``; ``nbatches = (n // batch_size
+ 1) * 3`` batches of ``random.sample(range(n), batch_size)``; the adjacency induced on the batch,
:58-69, 217-231), same fixed 3-layer ``SequentialGCN`` (:176-187), same loop and printed lines
(:241-306).  Every batch gets its own aggregation engine (partition of the induced matrix + its
own boundary maps, like :227-231) over the SAME kernels as the full-batch path; rows outside the
batch are empty rows of that engine, so their logits are zero exactly as in the reference.
All engines share one local row numbering (ascending global id): H and the labels are indexed once.
"""
from __future__ import annotations

import getopt
import os
import pickle
import random
import sys
import time

import numpy as np
import scipy.sparse as sparse
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

from . import PGCN as _P
from . import engine as _engine
from . import ingest as _ingest
from . import partition as _partition


def sample_adjacency_matrix(A, indices):
    """PGCN-Mini-batch.py:58-69: entries whose row AND column are in the batch."""
    keep = np.isin(A.row, indices) & np.isin(A.col, indices)
    return sparse.coo_matrix((A.data[keep], (A.row[keep], A.col[keep])), shape=A.shape)


class PGCN(nn.Module):
    """PGCN-Mini-batch.py:160-174: the layer takes the batch's adjacency at call time."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=False)

    def forward(self, A, H):
        H = _P.PSpMM.apply(A, H)
        H = _P._LinearNoBias.apply(H, self.linear.weight)
        return F.relu(H)


class SequentialGCN(nn.Module):
    """PGCN-Mini-batch.py:176-187."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.gcn1 = PGCN(in_features, out_features)
        self.gcn2 = PGCN(in_features, out_features)
        self.gcn3 = PGCN(in_features, out_features)

    def forward(self, A, H):
        return self.gcn3(A, self.gcn2(A, self.gcn1(A, H)))


def read_partvec(path):
    """``-p``: pickled list[int] (PGCN-Mini-batch.py:217-218, written by GPU/SHP/main.py:137-140);
    a one-line text part vector (the full-batch format) is accepted too."""
    try:
        with open(path, "rb") as f:
            return list(pickle.load(f))
    except (pickle.UnpicklingError, UnicodeDecodeError, EOFError, ValueError):
        return _partition.read_partvec(path)


def run(rank, size, nlayers, nfeatures, path_A, path_partvec, backend, batch_size):
    """PGCN-Mini-batch.py:201-306."""
    random_seed = 1
    np.random.seed(random_seed)
    random.seed(random_seed)
    _P.myrank, _P.world_size = rank, size
    if torch.cuda.is_available():
        device = torch.device(f'cuda:{rank % torch.cuda.device_count()}')
        torch.cuda.set_device(device)
    elif _P._kernel_provider is not None:
        device = torch.device('cpu')            # checker-backed provider injected by tests/
    else:
        raise _P._kernels._lib.PgcnError("no HIP device visible: refusing to run (no CPU fallback)")
    _P.device = device

    A = _ingest.mmread(path_A)
    partvec = read_partvec(path_partvec)
    n = A.shape[0]
    nbatches = (n // batch_size + 1) * 3

    exch = None
    if size > 1:
        if _P._exchanger is None:
            _P._exchanger = _engine.make_exchanger(rank, size, device, _P._exchange_impl)
        exch = _P._exchanger
    pv = torch.as_tensor(partvec, dtype=torch.int64)
    batches = []
    for _ in range(nbatches):
        batch_indices = np.array(random.sample(range(0, n), batch_size))
        bA = sample_adjacency_matrix(A, batch_indices)
        row, col, val = _P._coo_tensors(bA)
        part = _partition.build_partition(row, col, val, n, pv, rank, size, degree_sort=False)
        batches.append(_engine.AggregationEngine(part, _P._provider(), device, exch))
    _P.init_stats()

    owned = batches[0].part.owned.to(device)
    H = owned.to(torch.float32).unsqueeze(1).repeat(1, nfeatures).contiguous().requires_grad_(True)   # :236-238
    labels = owned % nfeatures                                                                        # :241

    model = SequentialGCN(nfeatures, nfeatures).to(device)
    _P.initiliaze_parameters(model)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)

    def one_batch(eng):
        logits = model(eng, H)
        loss = _P.local_loss(logits, labels, n)
        optimizer.zero_grad()
        loss.backward()
        _P.average_gradients(model)
        optimizer.step()
        return loss.detach()

    for eng in batches:                       # one untimed epoch, :252-268
        one_batch(eng)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    start = time.time()
    for epoch in range(4):
        loss_epoch = torch.ones((), device=device)          # the reference starts the sum at 1 (:273)
        for eng in batches:
            loss_epoch = loss_epoch + one_batch(eng)
        if size > 1:
            _P._all_reduce(loss_epoch)
        if rank == 0:
            print("Epoch {:05d} | Loss {:.4f}".format(epoch, loss_epoch), flush=True)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    elapsed = torch.tensor([time.time() - start], device=device)
    total_vol = torch.tensor(sum(e.stats["send_volume"] for e in batches), device=device)
    total_nmsg = torch.tensor(sum(e.stats["send_nmsg"] for e in batches), device=device)
    if size > 1:
        _P._all_reduce(elapsed, dist.ReduceOp.MAX)
        _P._all_reduce(total_vol)
        _P._all_reduce(total_nmsg)
    if rank == 0:
        print("Elapsed time {:.4f}".format(elapsed.item()), flush=True)
        print(f"total_vol: {total_vol} total_nmsg: {total_nmsg}", flush=True)
    return model


def init_process(rank, size, fn, nlayers, nfeatures, path_A, path_partvec, backend, batch_size):
    """PGCN-Mini-batch.py:309-321."""
    dist.init_process_group(backend, rank=rank, world_size=size)
    env_dict = {key: os.environ[key] for key in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE")}
    print(f"[{os.getpid()}] Initializing process group with: {env_dict}", flush=True)
    fn(rank, size, nlayers, nfeatures, path_A, path_partvec, backend, batch_size)
    if _P._exchanger is not None:
        _P._exchanger.close()
        _P._exchanger = None
    dist.destroy_process_group()


def main(argv):
    """PGCN-Mini-batch.py:324-354."""
    size = int(os.environ.get("SLURM_NPROCS", os.environ.get("WORLD_SIZE", "1")))
    rank = int(os.environ.get("SLURM_PROCID", os.environ.get("RANK", "0")))
    os.environ["RANK"] = str(rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    path_A = path_partvec = None
    backend, nlayers, nfeatures, batch_size = "nccl", 3, 128, 5
    try:
        opts, _ = getopt.getopt(argv, "a:p:b:s:l:f:n:", [])
    except getopt.GetoptError:
        print("a:p:b:", flush=True)
        sys.exit(2)
    for opt, arg in opts:
        if opt == '-a':
            path_A = arg
        elif opt == '-p':
            path_partvec = arg
        elif opt == '-b':
            backend = arg
        elif opt == '-s':
            size = int(arg)
        elif opt == '-l':
            nlayers = int(arg)
        elif opt == '-f':
            nfeatures = int(arg)
        elif opt == '-n':
            batch_size = int(arg)
    os.environ.setdefault("WORLD_SIZE", str(size))
    mp.set_start_method("spawn", force=True)
    p = mp.Process(target=init_process,
                   args=(rank, size, run, nlayers, nfeatures, path_A, path_partvec, backend, batch_size))
    p.start()
    p.join()
    if p.exitcode != 0:
        sys.exit(p.exitcode)


if __name__ == '__main__':
    main(sys.argv[1:])
