"""Aggregation engine: the per-layer ``A_p . H`` with boundary-row exchange.

This is the MI355X-native re-design of ``communicate_fgm`` + ``torch.sparse.mm``
(/root/reference/GPU/PGCN.py:85-134), structured like the CPU engine's overlap
(Parallel-GCN/main.c:238-299):

    forward   pack boundary rows        (gather kernel, compute stream)
              all-to-all-v round 0      (RCCL, comm stream)      ||  C  = A_loc    . H    (compute stream)
              all-to-all-v round 1                               ||  C += A_halo[0] . halo   (after round 0 landed)
                                                                     C += A_halo[1] . halo   (after round 1 landed)
    backward  P[r] = A_halo[r]^T . G    (partials for rows owned by peers, round by round)
              reverse all-to-all-v r    (comm stream, as soon as P[r] exists)  ||  P[r+1], dH = A_loc^T . G
              wait                                                   dH[send rows] += received partials (pattern SpMM)

Every peer's boundary list is cut into `rounds` parts (partition.Partition): each round is one
all-to-all-v over ALL peers (all xGMI links busy) on a contiguous sub-slab, and the halo pass of
round r overlaps the transfer of round r+1.

The received slab IS the halo panel: ``A_halo``'s column ids index it directly,
so the reference's n x f scratch ``X``, the ``H + X`` add and the n x n index
space disappear.  Unpack in backward ACCUMULATES (the reference assigns --
quirk Q3, SURVEY 8a -- which drops partial sums for P >= 3).
"""
from __future__ import annotations

import ctypes
import os
import dataclasses
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import _lib
from .partition import Partition


# --------------------------------------------------------------------------
# exchangers


class TorchDistExchanger:
    """all-to-all-v through torch.distributed (backend 'nccl' == RCCL, or 'gloo').

    With gloo and device tensors the slabs are staged through host memory; this is
    the transport used by the CPU multi-process tests and by ``-b gloo``."""

    name = "torch.distributed"

    def __init__(self, rank: int, size: int, group=None):
        self.rank, self.size, self.group = rank, size, group
        self.backend = dist.get_backend(group)

    def alltoallv(self, send: torch.Tensor, send_off: List[int], recv: torch.Tensor,
                  recv_off: List[int], f: int) -> None:
        in_split = [send_off[q + 1] - send_off[q] for q in range(self.size)]
        out_split = [recv_off[q + 1] - recv_off[q] for q in range(self.size)]
        s = send[:send_off[-1]]
        r = recv[:recv_off[-1]]
        if self.backend == "gloo" and s.is_cuda:
            sh = s.cpu()
            rh = torch.empty(r.shape, dtype=r.dtype)
            dist.all_to_all_single(rh, sh, out_split, in_split, group=self.group)
            r.copy_(rh)
        else:
            dist.all_to_all_single(r, s, out_split, in_split, group=self.group)

    def allreduce_sum(self, buf: torch.Tensor) -> None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    def close(self):
        pass


class RcclExchanger:
    """all-to-all-v through libpgcn_hip.so's own RCCL communicator (C ABI).

    The ncclUniqueId is created on rank 0 and broadcast through the already
    initialised torch.distributed process group (any backend)."""

    name = "rccl-capi"

    def __init__(self, rank: int, size: int, device: torch.device, group=None):
        self.lib = _lib.lib()
        self.rank, self.size, self.device = rank, size, device
        box = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            _lib.check(self.lib.pgcn_comm_unique_id(buf), "pgcn_comm_unique_id")
            box[0] = bytes(buf.raw)
        dist.broadcast_object_list(box, src=0, group=group)
        self._id = ctypes.create_string_buffer(box[0], 128)
        comm = ctypes.c_void_p()
        torch.cuda.set_device(device)
        _lib.check(self.lib.pgcn_comm_init(ctypes.byref(comm), self._id, size, rank), "pgcn_comm_init")
        self.comm = comm
        self._off_t = ctypes.c_int64 * (size + 1)

    def alltoallv(self, send: torch.Tensor, send_off: List[int], recv: torch.Tensor,
                  recv_off: List[int], f: int) -> None:
        so, ro = self._off_t(*send_off), self._off_t(*recv_off)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.pgcn_exchange_alltoallv_f32(
            self.comm, send.data_ptr() if send.numel() else None, so,
            recv.data_ptr() if recv.numel() else None, ro, f, stream), "pgcn_exchange_alltoallv_f32")

    def allreduce_sum(self, buf: torch.Tensor) -> None:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self.lib.pgcn_allreduce_sum_f32(self.comm, buf.data_ptr(), buf.numel(), stream),
                   "pgcn_allreduce_sum_f32")

    def close(self):
        if self.comm:
            self.lib.pgcn_comm_destroy(self.comm)
            self.comm = None


SELFTEST_TIMEOUT_S = float(os.environ.get("PGCN_SELFTEST_TIMEOUT", "90"))


def _wait_or_die(device: torch.device, what: str, timeout_s: float = None, on_timeout=None, poll_s: float = 0.002) -> None:
    """Wait for everything queued on ``device`` with a deadline: a collective whose peers never arrive (a mismatched
    send / receive pair, a dead link) spins on the GPU for ever -- the process then says what it was doing and
    exits with status 3 instead of hanging the launcher until its own limit."""
    import sys
    import time
    timeout_s = SELFTEST_TIMEOUT_S if timeout_s is None else timeout_s
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    t0 = time.time()
    while not ev.query():
        if time.time() - t0 > timeout_s:
            sys.stderr.write("pgcn: %s did not complete within %.0f s on %s -- the collective hangs (peer never arrived); "
                             "PGCN_EXCHANGE=torch selects torch.distributed's all_to_all_single instead of the C-ABI "
                             "communicator\n" % (what, timeout_s, device))
            sys.stderr.flush()
            if on_timeout is not None:          # (the caller's way out, e.g. bench.py printing the line it already has; must not return)
                on_timeout(what)
            os._exit(3)
        time.sleep(poll_s)


def exchange_selftest(ex, rank: int, size: int, device: torch.device, group=None, f: int = 4, rows: int = 3):
    """One all-to-all-v with NON-EMPTY, unequal segments (1 + (p + q) % rows rows between ranks p and q) through ``ex``,
    checked against what every peer must have sent: value = 1000 * sender + position in the sender's slab.  Device
    exchangers are waited for with a deadline.  Returns True / False (this rank's view)."""
    send_off, recv_off = [0], [0]
    for q in range(size):
        send_off.append(send_off[-1] + (0 if q == rank else 1 + (rank + q) % rows))
        recv_off.append(recv_off[-1] + (0 if q == rank else 1 + (rank + q) % rows))
    send = (torch.arange(send_off[-1] * f, device=device, dtype=torch.float32) + 1000 * rank).view(-1, f)
    got = torch.full((max(recv_off[-1], 1), f), -1.0, device=device)
    ex.alltoallv(send, send_off, got, recv_off, f)
    if torch.device(device).type == "cuda":
        _wait_or_die(device, "the exchanger self-test (%s, %d ranks)" % (ex.name, size))
    want = torch.full_like(got, -1.0)
    for q in range(size):
        if q == rank:
            continue
        # what q holds for me: its segment for target `rank` starts after its segments for targets < rank
        qoff = sum(0 if t == q else 1 + (q + t) % rows for t in range(rank))
        nrow = 1 + (rank + q) % rows
        seg = (torch.arange(qoff * f, (qoff + nrow) * f, device=device, dtype=torch.float32) + 1000 * q).view(-1, f)
        want[recv_off[q]:recv_off[q + 1]] = seg
    return bool(torch.equal(got[:recv_off[-1]], want[:recv_off[-1]]))


def make_exchanger(rank: int, size: int, device: torch.device, impl: str = "auto", group=None):
    """Pick the boundary-row transport.  'rccl' = libpgcn_hip.so's own communicator (C ABI),
    'torch' = torch.distributed all_to_all_single (RCCL under backend nccl, gloo otherwise).
    'auto' (or $PGCN_EXCHANGE) uses the C-ABI communicator on GPUs under nccl after a self-test with non-empty
    segments (both are RCCL: this is a choice of API, not a CPU fallback).  Whatever is chosen has passed
    ``exchange_selftest`` on every rank before it is returned; ``.selftest`` records what happened."""
    device = torch.device(device)
    backend = dist.get_backend(group)
    if impl == "auto":
        impl = os.environ.get("PGCN_EXCHANGE", "auto")

    def agreed(flag: float) -> bool:            # every rank takes the same decision
        ok = torch.tensor([flag], device=device if backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        return float(ok) == 1.0

    record = {}
    if not (impl == "torch" or device.type != "cuda" or backend != "nccl"):
        ex, good = None, 0.0
        try:
            ex = RcclExchanger(rank, size, device, group)
            good = 1.0 if exchange_selftest(ex, rank, size, device, group) else 0.0
        except _lib.PgcnError as e:
            record["rccl-capi_error"] = str(e)[:200]
        record["rccl-capi"] = bool(good)
        if agreed(good):
            ex.selftest = record
            return ex
        if ex is not None:
            ex.close()
        if impl == "rccl":
            raise _lib.PgcnError("C-ABI RCCL exchanger failed its self-test: %r" % (record,))
    ex = TorchDistExchanger(rank, size, group)
    good = 1.0 if (size == 1 or exchange_selftest(ex, rank, size, device, group)) else 0.0
    record["torch.distributed"] = bool(good)
    if not agreed(good):
        raise _lib.PgcnError("no boundary-row transport passed its self-test on every rank: %r" % (record,))
    ex.selftest = record
    return ex


# --------------------------------------------------------------------------


XGMI_LINK_GBS = 153.0      # one xGMI link of an MI355X, per direction (the bound of a round is max_q(s_pq f 4) / 153 GB/s)


class ExchangeProbe:
    """What the boundary exchange of a rank costs and how much of it the compute stream SEES (r06; bench.py's N > 1 line).
    Switched on around an eager timed region: every round of every exchange is bracketed by HIP events on the stream that runs
    it (the comm stream under overlap), every wait of the compute stream for a round by a pair of events on the compute stream
    (their distance = the time the aggregation stood still for that round = the EXPOSED part of the transfer), and the fused
    gradient all-reduce likewise.  Over a host-staged transport (gloo) the rounds are timed with the wall clock instead.
    ``summary()`` reduces the records per (direction, round): bytes out / in, the largest single peer segment (what one xGMI link
    carries), mean ms, mean exposed ms, GB/s on that link and its fraction of the link rate."""

    def __init__(self, device: torch.device):
        self.device, self.on = torch.device(device), False
        self.rounds, self.waits, self.reduces = [], [], []

    def clear(self):
        self.rounds, self.waits, self.reduces = [], [], []

    @staticmethod
    def _ms(x):
        a, b = x
        if isinstance(a, float):
            return 1e3 * (b - a)
        return a.elapsed_time(b)

    def summary(self, link_gbs: float = XGMI_LINK_GBS):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        out = {}
        keys = sorted({(t, r) for (t, r, *_x) in self.rounds})
        for tag, r in keys:
            recs = [x for x in self.rounds if x[0] == tag and x[1] == r]
            ms = [self._ms(x[5]) for x in recs]
            ex = [self._ms(x[2]) for x in self.waits if x[0] == tag and x[1] == r]
            _, _, b_out, b_in, b_peer, _ = recs[-1]
            mean_ms = sum(ms) / len(ms)
            gbs = b_peer / (mean_ms * 1e-3) / 1e9 if mean_ms > 0 else 0.0
            out.setdefault(tag, []).append({
                "round": r, "calls": len(recs), "bytes_out": b_out, "bytes_in": b_in, "max_peer_bytes": b_peer,
                "ms": mean_ms, "exposed_ms": (sum(ex) / len(ex)) if ex else mean_ms,
                "GBs_per_link": gbs, "frac_of_%dGBs" % int(link_gbs): gbs / link_gbs,
                "bound_ms_at_link_rate": b_peer / (link_gbs * 1e9) * 1e3})
        if self.reduces:
            ms = [self._ms(x[1]) for x in self.reduces]
            ex = [self._ms(x[2]) for x in self.reduces]
            out["allreduce"] = {"calls": len(ms), "bytes": self.reduces[-1][0], "ms": sum(ms) / len(ms), "exposed_ms": sum(ex) / len(ex)}
        return out


class BoundaryExchange:
    """Rank p's boundary-row slabs and the round-wise all-to-all-v over them (shared by the GCN
    aggregation engine and the GAT engine)."""

    def __init__(self, part: Partition, kernels, device: torch.device, exchanger=None,
                 overlap: Optional[bool] = None):
        self.part = part
        self.k = kernels
        self.device = torch.device(device)
        self.rank, self.size = part.rank, part.size
        self.exch = exchanger
        if self.size > 1 and exchanger is None:
            raise ValueError("a multi-rank partition needs an exchanger")
        self.send_idx = part.send_idx.to(self.device)
        self.unpack = [kernels.prepare(u, pattern_only=True) for u in part.unpack]
        self.rounds = part.rounds
        self.round_send_off = [list(o) for o in part.round_send_off]
        self.round_recv_off = [list(o) for o in part.round_recv_off]
        self.n_local, self.n_halo, self.n_send = part.n_local, part.n_halo, part.n_send
        self._buf: Dict = {}
        self.on_gpu = self.device.type == "cuda"
        if overlap is None:
            overlap = os.environ.get("PGCN_OVERLAP", "1") != "0"
        self.overlap = bool(overlap and self.on_gpu and self.size > 1)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.overlap else None
        # PGCN.py:78-83 counters (rows and messages); host integers -> no device kernels
        self.stats = {"send_volume": 0, "recv_volume": 0, "send_nmsg": 0, "recv_nmsg": 0}
        self.probe: Optional[ExchangeProbe] = None     # bench.py attaches one for its eager timed region

    # ------------------------------------------------------------------
    def _probing(self) -> Optional[ExchangeProbe]:
        p = self.probe
        return p if (p is not None and p.on) else None

    def _host_staged(self) -> bool:
        return (not self.on_gpu) or getattr(self.exch, "backend", "") == "gloo"

    def _round_bytes(self, src_off, dst_off, f):
        seg = lambda off, q: (off[q + 1] - off[q]) * f * 4
        peers = [q for q in range(self.size) if q != self.rank]
        peer = max([max(seg(src_off, q), seg(dst_off, q)) for q in peers] or [0])
        return (src_off[-1] - src_off[0]) * f * 4, (dst_off[-1] - dst_off[0]) * f * 4, peer

    def _timed(self, stream, fn):
        """Run fn() bracketed for the probe: HIP events on `stream`, or the wall clock over a host-staged transport."""
        if self._host_staged():
            import time
            if self.on_gpu:
                torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            fn()
            if self.on_gpu:
                torch.cuda.synchronize(self.device)
            return (t0, time.perf_counter())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fn()
        e1.record(stream)
        return (e0, e1)

    def _slab(self, name: str, rows: int, f: int) -> torch.Tensor:
        key = (name, f)
        t = self._buf.get(key)
        if t is None or t.shape[0] < rows:
            # zeros, not empty: a slab row that has not been received yet may sit in a staged panel of the
            # tiled kernels (multiplied by structural zeros) -- stale finite data is harmless, NaN bit patterns
            # of fresh memory would send the MFMA tiles down their exact (slow) path
            t = torch.zeros((max(rows, 1), f), dtype=torch.float32, device=self.device)
            self._buf[key] = t
        return t

    def _count(self, rows_out: int, rows_in: int) -> None:
        self.stats["send_volume"] += rows_out
        self.stats["recv_volume"] += rows_in
        self.stats["send_nmsg"] += self.size - 1   # PGCN.py:106 counts every peer, empty or not
        self.stats["recv_nmsg"] += self.size - 1

    def _exchange_round(self, src, src_off, dst, dst_off, f):
        """One all-to-all-v on the sub-slabs [off[0], off[size]) of src / dst (current stream)."""
        a0, a1 = src_off[0], src_off[-1]
        b0, b1 = dst_off[0], dst_off[-1]
        self.exch.alltoallv(src[a0:a1], [o - a0 for o in src_off], dst[b0:b1], [o - b0 for o in dst_off], f)

    def _exchange_all(self, src, src_offs, dst, dst_offs, f, ready=None, tag="forward"):
        """All rounds of one boundary-row exchange.  ``ready[r]`` (optional) is an event after which
        round r's source rows exist.  Returns one waiter per round (call it on the compute stream
        before touching that round's destination rows).  ``tag`` names the exchange for an attached ExchangeProbe."""
        self._count(src_offs[-1][-1], dst_offs[-1][-1])
        probe = self._probing()
        if not self.overlap:
            for r in range(self.rounds):
                if probe is None:
                    self._exchange_round(src, src_offs[r], dst, dst_offs[r], f)
                else:
                    cur = torch.cuda.current_stream(self.device) if self.on_gpu else None
                    span = self._timed(cur, lambda: self._exchange_round(src, src_offs[r], dst, dst_offs[r], f))
                    probe.rounds.append((tag, r) + self._round_bytes(src_offs[r], dst_offs[r], f) + (span,))
                    probe.waits.append((tag, r, span))           # nothing overlaps it: all of it is exposed
            return [lambda: None] * self.rounds
        main = torch.cuda.current_stream(self.device)
        if ready is None:
            ev = torch.cuda.Event()
            ev.record(main)
            ready = [ev] * self.rounds
        waiters = []
        with torch.cuda.stream(self.comm_stream):
            for r in range(self.rounds):
                self.comm_stream.wait_event(ready[r])
                if probe is None:
                    self._exchange_round(src, src_offs[r], dst, dst_offs[r], f)
                else:
                    span = self._timed(self.comm_stream, lambda: self._exchange_round(src, src_offs[r], dst, dst_offs[r], f))
                    probe.rounds.append((tag, r) + self._round_bytes(src_offs[r], dst_offs[r], f) + (span,))
                done = torch.cuda.Event()
                done.record(self.comm_stream)
                if probe is None or self._host_staged():
                    waiters.append(lambda d=done: main.wait_event(d))
                else:
                    def wait(d=done, r=r):
                        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        w0.record(main)
                        main.wait_event(d)
                        w1.record(main)
                        probe.waits.append((tag, r, (w0, w1)))
                    waiters.append(wait)
        return waiters

    def allreduce_sum(self, buf: torch.Tensor) -> None:
        """Sum ``buf`` (fp32, contiguous) over the ranks through the exchange transport.  With the
        comm stream on, the collective is issued THERE (after the producers of ``buf`` on the compute
        stream) so that every operation of the communicator is ordered on one stream; the compute
        stream waits for its completion."""
        if self.size == 1:
            return
        if not buf.is_contiguous():
            raise ValueError("allreduce_sum needs a contiguous buffer")
        if buf.is_cuda and getattr(self.exch, "backend", "") == "gloo":
            def staged():
                h = buf.cpu()
                self.exch.allreduce_sum(h)
                buf.copy_(h)
            probe = self._probing()
            if probe is None:
                staged()
            else:
                span = self._timed(None, staged)
                probe.reduces.append((buf.numel() * buf.element_size(), span, span))
            return
        probe = self._probing()
        nbytes = buf.numel() * buf.element_size()
        if not self.overlap:
            if probe is None:
                self.exch.allreduce_sum(buf)
            else:
                span = self._timed(torch.cuda.current_stream(self.device) if self.on_gpu else None, lambda: self.exch.allreduce_sum(buf))
                probe.reduces.append((nbytes, span, span))
            return
        main = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            if probe is None:
                self.exch.allreduce_sum(buf)
            else:
                span = self._timed(self.comm_stream, lambda: self.exch.allreduce_sum(buf))
            done = torch.cuda.Event()
            done.record(self.comm_stream)
        buf.record_stream(self.comm_stream)
        if probe is None:
            main.wait_event(done)
        else:
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record(main)
            main.wait_event(done)
            w1.record(main)
            probe.reduces.append((nbytes, span, (w0, w1)))


class AggregationEngine(BoundaryExchange):
    """Owns rank p's device-resident pieces and runs the aggregation forward/backward."""

    def __init__(self, part: Partition, kernels, device: torch.device, exchanger=None,
                 overlap: Optional[bool] = None):
        super().__init__(part, kernels, device, exchanger, overlap)
        self.A_loc = kernels.prepare(part.A_loc)
        self.A_halo = [kernels.prepare(a) for a in part.A_halo]
        if part.A_loc_T is part.A_loc:
            # one rank, symmetric A (partition._finish_partition checked it bit for bit): the SAME device structures serve the backward;
            # a second handle (own bindings and work-space, shared index / value / tile arrays) keeps forward and backward launches apart
            # for whoever times them by operand
            self.A_loc_T = dataclasses.replace(self.A_loc, ws=None, retired_ws=[], launch_cache={}) \
                if dataclasses.is_dataclass(self.A_loc) else self.A_loc
        else:
            self.A_loc_T = kernels.prepare(part.A_loc_T) if part.A_loc_T is not None else None
        self.A_halo_T = [kernels.prepare(a) for a in part.A_halo_T]

    # ------------------------------------------------------------------
    def forward(self, H: torch.Tensor) -> torch.Tensor:
        """AH = A_p . [H ; halo]   (H: n_local x f, owned rows only)."""
        if H.shape[0] != self.n_local:
            raise ValueError("H must hold exactly the %d owned rows" % self.n_local)
        H = H.contiguous()
        f = H.shape[1]
        C = torch.empty((self.n_local, f), dtype=torch.float32, device=self.device)
        if self.size == 1:
            return self.k.spmm(self.A_loc, H, C)
        send = self._slab("send", self.n_send, f)
        halo = self._slab("halo", self.n_halo, f)
        self.k.gather_rows(H, self.send_idx, send)
        waits = self._exchange_all(send, self.round_send_off, halo, self.round_recv_off, f, tag="forward")
        self.k.spmm(self.A_loc, H, C)            # overlaps the exchange (main.c:271)
        for r in range(self.rounds):
            waits[r]()
            self.k.spmm(self.A_halo[r], halo, C, accumulate=True)   # main.c:295; overlaps round r+1
        return C

    def backward(self, G: torch.Tensor) -> torch.Tensor:
        """dH = (A^T . G)[owned rows], partial sums returned to their owners and ADDED."""
        if self.A_loc_T is None:
            raise RuntimeError("partition was built without the transposed pieces")
        G = G.contiguous()
        f = G.shape[1]
        dH = torch.empty((self.n_local, f), dtype=torch.float32, device=self.device)
        if self.size == 1:
            return self.k.spmm(self.A_loc_T, G, dH)
        partial = self._slab("halo", self.n_halo, f)
        back = self._slab("send", self.n_send, f)
        ready = []
        for r in range(self.rounds):             # round r's partial rows, then hand them to the wire
            b0, b1 = self.round_recv_off[r][0], self.round_recv_off[r][-1]
            self.k.spmm(self.A_halo_T[r], G, partial[b0:b1])
            if self.overlap:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                ready.append(ev)
        waits = self._exchange_all(partial, self.round_recv_off, back, self.round_send_off, f,
                                   ready if self.overlap else None, tag="backward")
        self.k.spmm(self.A_loc_T, G, dH)
        for r in range(self.rounds):
            waits[r]()
            # dH[row] += the partial rows that came back for it in this round: one pattern SpMM
            # (deterministic; the reference ASSIGNS here and loses partials for P >= 3, quirk Q3)
            self.k.spmm(self.unpack[r], back, dH, accumulate=True)
        return dH

    def forward_symmetric_backward(self, G: torch.Tensor) -> torch.Tensor:
        """Parallel-GCN's backward (main.c:343-404): exchange rows of G with the FORWARD
        maps and apply A (valid when A = A^T)."""
        return self.forward(G)

    # ------------------------------------------------------------------
    def alg_bytes_forward(self, f: int) -> int:
        """Compulsory HBM bytes of one forward aggregation on this rank."""
        b = self.A_loc.alg_bytes(f)
        if self.size > 1:
            b += sum(8 * a.nnz + 8 * (a.nrows + 1) for a in self.A_halo) + 4 * f * self.n_halo
            b += 2 * 4 * f * self.n_send   # pack: read rows + write slab
        return b
