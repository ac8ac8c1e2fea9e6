"""The aggregation path driven with the CPU engine's training semantics.

Mirror of ``GCN()`` in /root/reference/Parallel-GCN/main.c:166-454 (sigmoid layers,
the loss of :318-323, the output gradient of :325-335, backward with ``A`` -- not
``A^T`` -- :376, dW Allreduce SUM :425, SGD :430), with both aggregations per layer
executed by the HIP engine (``AggregationEngine.forward``).  Dense algebra (n_p x f by
f x f) is plain torch -- plumbing around the graded path.  This is the form the
north-star parity clause names ("outputs match the reference Parallel-GCN CPU path").
"""
from __future__ import annotations

import getopt
import os
import sys
import time
from typing import Dict, Sequence

import numpy as np
import torch


def _sigmoid(x):
    return 1 / (1 + torch.exp(-x))          # main.c:79-81


def train(engine, d: Sequence[int], W: Dict[int, torch.Tensor], H0: torch.Tensor, Y: torch.Tensor,
          Ymask: torch.Tensor, epochs: int = 3, alpha: float = 0.01, allreduce=None):
    """``H0, Y, Ymask`` hold OWNED rows only.  ``allreduce(t)`` sums a tensor over ranks in
    place (None for a single rank).  Returns (err per epoch, W, output H_{L-1})."""
    L = len(d) - 1
    n = d[0]
    W = {l: W[l].clone() for l in range(1, L)}
    H = {0: H0}
    Z = {}
    errs = []
    Ym = Ymask.bool()
    for _ in range(epochs):                                        # main.c:231
        for l in range(1, L):                                      # :233
            AH = engine.forward(H[l - 1])                          # :238-299
            Z[l] = AH @ W[l]                                       # :303
            H[l] = _sigmoid(Z[l])                                  # :308
        P = H[L - 1]
        T = torch.where(Ym, (-1.0 * Y.double() * torch.log(P.double())).float(), P)   # :70-73, 318
        err = T.sum().reshape(1)                                   # :320
        if allreduce is not None:
            allreduce(err)                                         # :321
        errs.append(float(err))
        D = torch.where(Ym, P - Y, P) / (P * (1 - P))              # :325-328
        s = _sigmoid(Z[L - 1])
        G = (D * (s * (1 - s))) / float(n)                         # :330-335
        for l in range(L - 1, 0, -1):                              # :338
            AG = engine.forward_symmetric_backward(G)              # :343-404 (A, relies on A = A^T)
            if l != 1:
                s = _sigmoid(Z[l - 1])
                Gn = (AG @ W[l].t()) * (s * (1 - s))               # :407-410
            dW = H[l - 1].t() @ AG                                 # :417
            if allreduce is not None:
                allreduce(dW)                                      # :425
            W[l] = W[l] - alpha * dW                               # :430
            if l != 1:
                G = Gn
    return errs, W, H[L - 1]


def glibc_rand(seed: int, count: int) -> np.ndarray:
    """``srand(seed)`` followed by ``count`` calls of ``rand()`` as glibc computes them (the TYPE_3 additive-feedback
    generator of random_r.c: 31 words from a Lehmer sequence, r[i] = r[i-31] + r[i-3], 310 outputs discarded, top 31
    bits returned) -- the stream main.c:555,569-571 draws its weights from."""
    seed = int(seed) & 0xffffffff
    word = (seed - (1 << 32) if seed >= (1 << 31) else seed) or 1          # the int32 the state starts from; 0 -> 1
    r = [word]
    for _ in range(30):
        sign = -1 if r[-1] < 0 else 1
        hi, lo = sign * (abs(r[-1]) // 127773), sign * (abs(r[-1]) % 127773)   # C division: towards zero
        w = 16807 * lo - 2836 * hi
        r.append(w + 2147483647 if w < 0 else w)
    r = [v & 0xffffffff for v in r]
    r += r[0:3]
    for i in range(34, 344 + count):
        r.append((r[i - 31] + r[i - 3]) & 0xffffffff)
    return np.asarray(r[344:], dtype=np.int64) >> 1


def init_weights(d: Sequence[int], seed: int = 0, stream: str = "numpy") -> Dict[int, np.ndarray]:
    """main.c:578-595: W[l] ~ U(-sd, sd), sd = sqrt(6 / (ni + no)).  The reference seeds rand() with time(NULL) on
    every rank (unreproducible, replicas may differ).  ``stream="numpy"``: a seeded numpy stream;
    ``stream="glibc"``: the reference's own arithmetic on glibc's ``srand(seed)`` stream -- bit for bit the weights
    the reference binary draws when its clock reads ``seed`` (get_random, main.c:93-96: float division by RAND_MAX,
    ``x + (y - x) * r`` in float; layers, rows and columns in the order of main.c:566-573)."""
    L = len(d) - 1
    if stream == "glibc":
        sizes = [d[l] * d[l + 1] for l in range(1, L)]
        draws = glibc_rand(seed, sum(sizes)).astype(np.float32) / np.float32(2147483647)
        W, at = {}, 0
        for l in range(1, L):
            sd = np.float32(np.sqrt(6.0 / float(np.float32(d[l] + d[l + 1]))))
            r = draws[at:at + sizes[l - 1]].reshape(d[l], d[l + 1])
            W[l] = (-sd) + (sd - (-sd)) * r
            at += sizes[l - 1]
        return W
    if stream != "numpy":
        raise ValueError("unknown weight stream %r" % (stream,))
    rng = np.random.default_rng(seed)
    return {l: ((rng.random((d[l], d[l + 1]), dtype=np.float32) * 2 - 1) *
                np.float32(np.sqrt(6.0 / (d[l] + d[l + 1])))) for l in range(1, L)}


def main(argv, kernels=None, out=None):
    """Drop-in for the CPU engine's command line  ``grbgcn -p DATA_DIR -c CONFIG -t nthreads``
    (Parallel-GCN/main.c:99-164, README.md:57-81) on the HIP engine.  Prints what rank 0 of the
    reference prints: the config echo (main.c:699-704), ``err:%g`` per epoch (:323),
    ``time : %f secs`` (:445) and the 8 communication statistics (:519-522).  Rank / world size
    from the launcher's RANK / WORLD_SIZE (mpirun in the reference); WORLD_SIZE must be 1 or the
    number of parts in DATA_DIR.  Returns (errs, W, output rows, partition)."""
    import torch.distributed as dist

    from . import engine as _engine
    from . import kernels as _kernels
    from . import pargcn_io as _io
    from . import partition as _partition
    out = out or sys.stdout
    path, config_path = None, None
    opts, _ = getopt.getopt(argv, "p:t:c:")
    for o, a in opts:
        if o == "-p":
            path = a
        elif o == "-c":
            config_path = a
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    own_group = False
    if world > 1 and not dist.is_initialized():      # the reference is started by mpirun; here: torchrun
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(os.environ.get("PGCN_BACKEND", "nccl"), rank=rank, world_size=world)
        own_group = True
    prob = _io.load_directory(path, config_path)
    L, d, n = prob["L"], prob["d"], prob["d"][0]
    if world not in (1, prob["k"]):
        raise ValueError("%s holds %d parts; run with 1 or %d ranks" % (path, prob["k"], prob["k"]))
    if rank == 0:
        print("nlayers:%d" % L, file=out)
        print(" ".join(str(x) for x in d) + " ", file=out)
    if kernels is None:
        if not torch.cuda.is_available():
            raise _kernels._lib.PgcnError("no HIP device visible: refusing to run (no CPU fallback)")
        dev = torch.device("cuda:%d" % (int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count()))
        torch.cuda.set_device(dev)
        kernels = _kernels.HipKernels(dev)
    else:
        dev = torch.device("cpu")
    A = prob["A"]
    # The reference multiplies by what its conn.* files deliver, not by what the pattern needs (main.c:250,293-295);
    # GCN-HP derives those lists from the sender's rows (GCN-HP/main.cpp:147-176), so on an unsymmetric pattern some
    # entries refer to rows that never arrive and contribute nothing.  Same inputs, same numbers: drop them here too.
    keep = _io.delivered_rows(prob["conn"], prob["part"], prob["k"])[prob["part"][A.row], A.col]
    if not keep.all():
        if rank == 0:
            print("pargcn: %d of %d stored entries refer to rows that the conn.* files of %s never deliver (unsymmetric "
                  "pattern); the reference engine ignores them and so does this run" % (int((~keep).sum()), A.nnz, path),
                  file=sys.stderr)
        A = type(A)((A.data[keep], (A.row[keep], A.col[keep])), shape=A.shape)
    partvec = torch.from_numpy(prob["part"] if world > 1 else np.zeros(n, dtype=np.int64))
    part = _partition.build_partition(torch.from_numpy(A.row.astype(np.int64)), torch.from_numpy(A.col.astype(np.int64)),
                                      torch.from_numpy(A.data.astype(np.float32)), n, partvec, rank, world)
    exch = _engine.make_exchanger(rank, world, dev) if world > 1 else None
    eng = _engine.AggregationEngine(part, kernels, dev, exch)
    own = part.owned.numpy()
    seed = os.environ.get("PGCN_SEED", "0")                # "7": numpy stream; "glibc:7": the reference's srand(7) stream
    stream, seed = ("glibc", seed[6:]) if seed.startswith("glibc:") else ("numpy", seed)
    W = {l: torch.from_numpy(w).to(dev) for l, w in init_weights(d, int(seed), stream).items()}
    H0 = torch.ones((own.size, d[1]), device=dev)                                  # main.c:650-685
    Y = torch.from_numpy(prob["Y"][own]).to(dev)
    Ym = torch.from_numpy(prob["Ymask"][own]).to(dev)
    allreduce = None
    if world > 1:
        def allreduce(t):          # main.c:321,425 MPI_Allreduce SUM -- on the engine's own transport / stream
            c = t.contiguous()
            eng.allreduce_sum(c)
            if c is not t:
                t.copy_(c)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    tic = time.time()
    errs, Wn, Hout = train(eng, d, W, H0, Y, Ym, epochs=3, alpha=0.01, allreduce=allreduce)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    elapsed = torch.tensor([time.time() - tic], dtype=torch.float64)
    # statistics(), main.c:506-524: volumes in SCALARS (rows x width, :264), per rank
    widths = [d[l] for l in range(1, L)] + [d[l + 1] for l in range(L - 1, 0, -1)]     # per epoch
    rows_out, rows_in = part.n_send, part.n_halo
    nz_t = int(torch.unique(part.send_owner).numel())      # targets I really send to (Hsend != NULL, main.c:247)
    nz_s = int(torch.unique(part.halo_owner).numel())      # sources (Hrecv != -1, main.c:239)
    st = torch.tensor([rows_out * sum(widths) * 3, rows_in * sum(widths) * 3, nz_t * len(widths) * 3,
                       nz_s * len(widths) * 3], dtype=torch.float64)
    if world > 1:
        # NCCL (= RCCL) process groups only move device tensors, gloo only host tensors here
        wire = dev if (dev.type == "cuda" and dist.get_backend() != "gloo") else torch.device("cpu")
        tot, mx, el = st.to(wire).clone(), st.to(wire).clone(), elapsed.to(wire).clone()   # (.to() on the same device aliases)
        dist.all_reduce(tot)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        tot, mx, elapsed = tot.cpu(), mx.cpu(), el.cpu()
    else:
        tot, mx = st, st
    if rank == 0:
        for e in errs:
            print("err:%g" % e, file=out)
        print("time : %f secs" % float(elapsed), file=out)
        print("%d %d %d %d %d %d %d %d" % (tot[0], tot[0] // world, mx[0], mx[1], tot[2], tot[2] // world,
                                           mx[2], mx[3]), file=out)
    if own_group:
        if exch is not None:
            exch.close()
        dist.destroy_process_group()
    return errs, Wn, Hout, part


if __name__ == "__main__":
    main(sys.argv[1:])
