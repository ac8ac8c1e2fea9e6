"""Device kernels of the aggregation path: thin typed wrappers over the C ABI.

``HipKernels`` is the ONLY kernel provider the product constructs.  It takes
PyTorch CUDA tensors (PyTorch = device memory + streams, i.e. plumbing), passes
raw device pointers and the current ``hipStream_t`` to ``libpgcn_hip.so`` and
never computes anything itself.  There is no CPU implementation in this package:
without a HIP device (or without the built library) construction raises.

The small ``Kernels`` protocol exists so that the host logic (exchange ordering,
PSpMM forward/backward structure, statistics) can be exercised by the CPU-only
multi-process ``gloo`` tests with a checker-backed provider that lives in
``tests/`` -- never in the product.
"""
from __future__ import annotations

import ctypes
import dataclasses
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib
from .partition import HostCSR
from .tuning import T as _T

DEFAULT_CHUNK = _T.spmm_chunk
DEFAULT_SMALL_ROW = _T.spmm_small_row
GROUP_MIN_ROW = _T.group_min_row


@dataclass
class DeviceCSR:
    """A CSR block resident in HBM plus its load-balancing plan."""
    nrows: int
    ncols: int
    nnz: int
    rowptr: torch.Tensor            # int64 [nrows+1]
    col: torch.Tensor               # int32 [nnz]
    val: Optional[torch.Tensor]     # fp32 [nnz] (None = pattern)
    row_map: Optional[torch.Tensor]  # int32 [nrows] or None
    tasks: Optional[torch.Tensor] = None  # int32 [ntasks,4]
    fix: Optional[torch.Tensor] = None    # int32 [nfix,4]
    ntasks: int = 0
    nfix: int = 0
    nslots: int = 0
    nslices: int = 1
    seg: Optional[object] = None          # ctypes int64[nslices+1] (host) task segment bounds
    ws: Optional[torch.Tensor] = None     # fp32 work-space for split rows (grown on demand)
    retired_ws: list = field(default_factory=list)   # outgrown work-spaces, kept allocated (see _bind_spmm)
    core: Optional["DeviceCore"] = None   # dense-tile part (LDS-tiled kernel)
    strip: Optional["DeviceStrip"] = None  # 512 x 128 strip tiles (LDS-staged, async pipeline)
    dense3: Optional["DeviceDense3"] = None  # densest 512 x 128 blocks (bf16 matrix cores, three planes)
    fix_all: Optional[torch.Tensor] = None    # int32 [nfix_all,4] combined fix list (core + gather slots)
    slot_ids: Optional[torch.Tensor] = None   # int32 slot lists of fix_all
    nslots_total: int = 0
    rows_wave: Optional[torch.Tensor] = None   # GAT kernels: int32 rows handled by one wave each ...
    rows_block: Optional[torch.Tensor] = None  # ... and by one 256-thread workgroup each (hub rows)
    slice_off: Optional[torch.Tensor] = None   # int32 [nrows, 9] offsets of a row's 8 col % 8 slices (XCD-sliced storage)
    rows_all: Optional[torch.Tensor] = None    # int32 all rows, longest first
    launch_cache: dict = None                 # bound C-ABI calls per (ldb, ldc, f, accumulate)

    def __post_init__(self):
        if self.launch_cache is None:
            self.launch_cache = {}

    def alg_bytes(self, f: int, n_cols_touched: Optional[int] = None, n_rows_out: Optional[int] = None) -> int:
        """Algorithmic (compulsory) HBM bytes of one SpMM, SURVEY 8(d):
        8.nnz + 8.(nrows+1) + 4.f.(cols touched) + 4.f.(rows written)."""
        nc = self.ncols if n_cols_touched is None else n_cols_touched
        nr = self.nrows if n_rows_out is None else n_rows_out
        return 8 * self.nnz + 8 * (self.nrows + 1) + 4 * f * nc + 4 * f * nr


@dataclass
class DeviceDense3:
    work: torch.Tensor
    blk_img: torch.Tensor
    vals3: torch.Tensor
    panel_list: torch.Tensor
    npieces: int
    npanels: int
    nnz: int
    image: Optional[torch.Tensor] = None      # uint8 work-space of the split panels (grown on demand, per width)
    retired: list = field(default_factory=list)   # outgrown images: a HIP graph captured earlier still launches into them


@dataclass
class DeviceGatBlocks:
    """The dense 512 x 128 blocks of an attention pattern (pgcn_gat_blocks.hip): the block structure of ``DeviceDense3`` with the PATTERN
    as one bit per position, and the slot lists that add a piece's partial rows to the outputs."""
    nrows: int
    ncols: int
    work: torch.Tensor            # int32 [npieces, 4]
    work_row0: torch.Tensor       # int32 [npieces] first matrix row of a piece
    blk_img: torch.Tensor
    bits: torch.Tensor            # int32 [nblocks, 8, 64, 4]
    panel_list: torch.Tensor
    fix: torch.Tensor             # int32 [rows with a slot, 4] {row, begin, count, 0}
    slot_ids: torch.Tensor
    npieces: int
    npanels: int
    nslots: int
    nnz: int
    image: Optional[torch.Tensor] = None
    ws: Optional[torch.Tensor] = None


@dataclass
class DeviceStrip:
    work: torch.Tensor
    rec: torch.Tensor
    pairs: torch.Tensor
    npieces: int
    nnz: int


@dataclass
class DeviceCore:
    work: torch.Tensor
    tile_panel: torch.Tensor
    tile_base: torch.Tensor
    seg_off: torch.Tensor
    ccol: torch.Tensor
    cval: torch.Tensor
    npieces: int
    nnz: int


def build_plan(rowptr_host: np.ndarray, chunk: int, slice_cnt: Optional[np.ndarray] = None,
               small_row: int = DEFAULT_SMALL_ROW, force: bool = False,
               row_flags: Optional[np.ndarray] = None, ngroups: int = 1,
               group_min_row: int = GROUP_MIN_ROW):
    """Host-side task list (pgcn_spmm_plan_host).  Returns (tasks, fix, nslots, seg) with
    numpy int32 arrays; tasks is None when the plan is trivial (unsliced and no row
    exceeds ``chunk``): the one-task-per-row kernel path needs no plan."""
    L = _lib.lib()
    rowptr_host = np.ascontiguousarray(rowptr_host, dtype=np.int64)
    nrows = rowptr_host.shape[0] - 1
    S = 1
    sc_ptr = None
    if slice_cnt is not None:
        slice_cnt = np.ascontiguousarray(slice_cnt, dtype=np.int32)
        S = slice_cnt.shape[1] // ngroups
        assert S * ngroups == slice_cnt.shape[1]
        sc_ptr = slice_cnt.ctypes.data
    rf_ptr = None
    if row_flags is not None:
        row_flags = np.ascontiguousarray(row_flags, dtype=np.uint8)
        rf_ptr = row_flags.ctypes.data
        force = True
    seg = (ctypes.c_int64 * (S + 1))()
    nt, nf, ns = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    plan_flags = 0x40000000 if (_T.spmm_affine_small and slice_cnt is not None) else 0     # PGCN_PLAN_AFFINE_SMALL
    _lib.check(L.pgcn_spmm_plan_host_ex(rowptr_host.ctypes.data, sc_ptr, rf_ptr, nrows, S, ngroups, group_min_row, chunk, small_row, plan_flags,
                                        None, 0, None, 0, seg, ctypes.byref(nt), ctypes.byref(nf), ctypes.byref(ns)),
               "pgcn_spmm_plan_host")
    if nf.value == 0 and S == 1 and not force:
        return None, None, 0, None
    tasks = np.empty((nt.value, 4), dtype=np.int32)
    fix = np.empty((max(nf.value, 1), 4), dtype=np.int32)
    _lib.check(L.pgcn_spmm_plan_host_ex(rowptr_host.ctypes.data, sc_ptr, rf_ptr, nrows, S, ngroups, group_min_row, chunk, small_row, plan_flags,
                                        tasks.ctypes.data, nt.value, fix.ctypes.data, nf.value, seg, ctypes.byref(nt),
                                        ctypes.byref(nf), ctypes.byref(ns)), "pgcn_spmm_plan_host")
    return tasks, fix[:nf.value], int(ns.value), seg


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class HipKernels:
    """libpgcn_hip.so on one MI355X.  Fails loudly when the device or library is missing."""

    name = "hip"

    def __init__(self, device: torch.device, xcd_swizzle: Optional[bool] = None, chunk: int = DEFAULT_CHUNK,
                 small_row: int = DEFAULT_SMALL_ROW):
        if not torch.cuda.is_available():
            raise _lib.PgcnError("HipKernels needs a HIP device: torch.cuda.is_available() is False "
                                 "(this package has no CPU fallback)")
        self.lib = _lib.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PgcnError("HipKernels needs a cuda (HIP) device, got %s" % device)
        if xcd_swizzle is None:
            xcd_swizzle = _T.xcd_swizzle
        self.base_flags = _lib.SPMM_XCD_SWIZZLE if xcd_swizzle else 0
        # feature passes of the gather kernels (tuning.fpass = "64": 64 features per pass, "0": whole rows; default
        # "auto": 64 where the operand panel (>= 96 MB) is several times the aggregate L2 and the gather part holds
        # >= 8 M entries -- whole graphs; a rank's shard of an 8-way run loses 15 % with them: tools/rank_probe.py, r02)
        self.fpass = _T.fpass
        self.sides = []                   # extra streams of tuning.lanes, made on first use
        self.single_lane = False          # bench.py's per-kernel split: everything on the current stream
        if self.fpass == "64":
            self.base_flags |= _lib.SPMM_FPASS64
        self.chunk = chunk
        self.small_row = small_row
        self.adaptive_chunk = _T.spmm_adaptive_chunk

    # -- data placement -------------------------------------------------
    def prepare(self, csr: HostCSR, pattern_only: bool = False, chunk: Optional[int] = None,
                small_row: Optional[int] = None) -> DeviceCSR:
        """``chunk`` / ``small_row``: plan parameters of THIS matrix when they differ from the provider's (the attention structures
        of the GAT path: tuning.gat_chunk / gat_small_row); arguments, not provider state, so two engines can be built at once."""
        dev = self.device
        rowptr_host = csr.rowptr.detach().cpu().numpy()
        sc = None if csr.slice_cnt is None else csr.slice_cnt.detach().cpu().numpy()
        rf = None if csr.row_flags is None else csr.row_flags.detach().cpu().numpy()
        # a task is a latency chain of chunk / 8 gather batches (~0.7 us each): on a small block (a rank's shard of an
        # 8-way run) a 1 024-entry task outlasts the rest of the kernel, so the chunk shrinks with the block --
        # entries / 8 192, between 64 and the configured chunk (r03: tools/probes_r03, emulated ranks)
        top = self.chunk if chunk is None else int(chunk)
        chunk = top
        if self.adaptive_chunk:
            e = max(int(csr.col.numel()), 1) // 8192
            chunk = min(top, max(64, 1 << max(e.bit_length() - 1, 0)))
        tasks, fix, nslots, seg = build_plan(rowptr_host, chunk, sc, self.small_row if small_row is None else int(small_row),
                                             force=csr.row_map is not None, row_flags=rf,
                                             ngroups=csr.ngroups)
        d = DeviceCSR(
            nrows=csr.nrows, ncols=csr.ncols, nnz=csr.nnz,
            rowptr=csr.rowptr.to(dev, torch.int64).contiguous(),
            col=csr.col.to(dev, torch.int32).contiguous(),
            val=None if pattern_only else csr.val.to(dev, torch.float32).contiguous(),
            row_map=None if csr.row_map is None else csr.row_map.to(dev, torch.int32).contiguous())
        if tasks is not None:
            d.tasks = torch.from_numpy(tasks).to(dev)
            d.fix = torch.from_numpy(fix).to(dev) if fix.shape[0] else None
            d.ntasks, d.nfix, d.nslots = tasks.shape[0], fix.shape[0], nslots
            d.nslices, d.seg = csr.nslices if sc is not None else 1, seg
        d.nslots_total = d.nslots
        if csr.core is not None or csr.strip is not None or csr.dense3 is not None:
            self._attach_core(d, csr, fix if tasks is not None else None)
        return d

    def _attach_core(self, d: DeviceCSR, csr: HostCSR, fix_rem: Optional[np.ndarray]) -> None:
        """Upload the tiled parts (LDS core, MFMA tiles) and build the per-row slot lists: a row's
        partial sums are its core pieces (in work order), its dense pieces, then its gather-kernel slots."""
        from .partition import CORE_TR, STRIP_TR
        dev = self.device
        hc, hs, h3 = csr.core, csr.strip, csr.dense3
        ns_rem = d.nslots
        pieces = []                                                # (first row, rows, first slot) of every piece
        ns_core = ns_strip = 0
        if hs is not None:
            work = hs.work.clone()
            work[:, 3] += ns_rem                                   # strip slots live behind the gather slots
            d.strip = DeviceStrip(work.to(dev).contiguous(), hs.rec.to(dev).contiguous(), hs.pairs.to(dev).contiguous(),
                                  hs.npieces, hs.nnz)
            w64 = work.cpu().to(torch.int64)
            pieces.append(torch.stack([w64[:, 0] * STRIP_TR, torch.full_like(w64[:, 0], STRIP_TR), w64[:, 3]], 1))
            ns_strip = hs.nslots
        if hc is not None:
            work = hc.work.clone()
            work[:, 3] += ns_rem + ns_strip                        # core slots behind those
            d.core = DeviceCore(work.to(dev).contiguous(), hc.tile_panel.to(dev), hc.tile_base.to(dev),
                                hc.seg_off.to(dev).contiguous(), hc.ccol.to(dev), hc.cval.to(dev),
                                hc.npieces, hc.nnz)
            w64 = work.cpu().to(torch.int64)
            pieces.append(torch.stack([w64[:, 0] * CORE_TR, torch.full_like(w64[:, 0], CORE_TR), w64[:, 3]], 1))
            ns_core = hc.nslots
        if h3 is not None:
            work = h3.work.clone()
            work[:, 3] += ns_rem + ns_strip + ns_core              # ... and the bf16 blocks last
            d.dense3 = DeviceDense3(work.to(dev).contiguous(), h3.blk_img.to(dev).contiguous(), h3.vals3.to(dev).contiguous(),
                                    h3.panel_list.to(dev).contiguous(), h3.npieces, int(h3.panel_list.numel()), h3.nnz)
            w64 = work.cpu().to(torch.int64)
            # (a block may start at any row and its band may end inside it: the piece's own first row and row count)
            pieces.append(torch.stack([h3.piece_row0.cpu().to(torch.int64), h3.piece_rows.cpu().to(torch.int64), w64[:, 3]], 1))
        wk = torch.cat(pieces)
        cntp = wk[:, 1]
        startp = torch.cumsum(cntp, 0) - cntp
        rit = torch.arange(int(cntp.sum()), dtype=torch.int64) - torch.repeat_interleave(startp, cntp)
        rows = torch.repeat_interleave(wk[:, 0], cntp) + rit
        slots = torch.repeat_interleave(wk[:, 2], cntp) + rit
        seq = torch.repeat_interleave(torch.arange(wk.shape[0], dtype=torch.int64), cntp)
        ok = rows < csr.nrows
        rows, slots, seq = rows[ok], slots[ok], seq[ok]
        if fix_rem is not None and fix_rem.shape[0]:
            fr = torch.from_numpy(np.ascontiguousarray(fix_rem)).to(torch.int64)
            cnt = fr[:, 2]
            r2 = torch.repeat_interleave(fr[:, 0], cnt)
            first = torch.repeat_interleave(fr[:, 1], cnt)
            start = torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
            s2 = first + (torch.arange(int(cnt.sum()), dtype=torch.int64) - start)
            q2 = torch.full_like(r2, 1 << 40) + s2
            rows, slots, seq = torch.cat([rows, r2]), torch.cat([slots, s2]), torch.cat([seq, q2])
        order = torch.argsort(rows * (1 << 42) + seq)
        rows, slots = rows[order], slots[order]
        urows, counts = torch.unique_consecutive(rows, return_counts=True)
        begin = torch.cumsum(counts, 0) - counts
        fix_all = torch.stack([urows, begin, counts, torch.zeros_like(urows)], 1).to(torch.int32)
        d.fix_all = fix_all.to(dev).contiguous()
        d.slot_ids = slots.to(torch.int32).to(dev).contiguous()
        d.nslots_total = ns_rem + ns_strip + ns_core + (h3.nslots if h3 is not None else 0)
        d.nnz = csr.nnz

    # -- kernels ----------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check_dense(self, t: torch.Tensor, rows: int, what: str):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise _lib.PgcnError("%s must be a row-major fp32 CUDA matrix" % what)
        if t.shape[0] < rows:
            raise _lib.PgcnError("%s has %d rows, need %d" % (what, t.shape[0], rows))

    def spmm(self, A: DeviceCSR, B: torch.Tensor, C: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
        """C (+)= A.B on the current stream.  C rows are addressed through A.row_map if set.

        The argument lists of the C-ABI calls only change in three pointers from call to call, so
        they are bound once per (leading dimensions, width, accumulate) and replayed: an epoch on
        8 GPUs is ~120 launches of ~30 us of device time each, the Python side must stay out of
        the way."""
        if not (B.is_cuda and C.is_cuda and B.dtype is torch.float32 and C.dtype is torch.float32):
            raise _lib.PgcnError("B and C must be fp32 CUDA matrices")
        if B.shape[0] < A.ncols or (A.row_map is None and C.shape[0] < A.nrows):
            raise _lib.PgcnError("B or C has too few rows")
        key = (B.stride(0), C.stride(0), B.shape[1], accumulate, C.shape[1], B.stride(1), C.stride(1))
        fn = A.launch_cache.get(key)
        if fn is None:
            fn = self._bind_spmm(A, B, C, accumulate)
            A.launch_cache[key] = fn
        fn(B, C)
        return C

    def _bind_spmm(self, A: DeviceCSR, B: torch.Tensor, C: torch.Tensor, accumulate: bool):
        f = B.shape[1]
        self._check_dense(B, A.ncols, "B")
        self._check_dense(C, 0, "C")
        if C.shape[1] != f:
            raise _lib.PgcnError("B and C widths differ")
        ldb, ldc = B.stride(0), C.stride(0)
        flags = self.base_flags | (_lib.SPMM_ACCUMULATE if accumulate else 0)
        if self.fpass == "auto" and f > 64 and A.ncols * f * 4 >= (96 << 20) and A.col.numel() >= 8_000_000:
            flags |= _lib.SPMM_FPASS64
        if (A.ncols + 1) * ldb * 4 < 2 ** 32:
            flags |= _lib.SPMM_OFFSETS32
        lib, check, stream = self.lib, _lib.check, self._stream
        if A.nrows == 0:
            return lambda B, C: None
        if A.nnz == 0 and A.core is None and A.strip is None and A.dense3 is None:   # nothing to launch: C = 0 (memset, plumbing) or C unchanged
            if accumulate:
                return lambda B, C: None
            if A.row_map is None:
                return lambda B, C: C[:A.nrows].zero_()
            rows = A.row_map.long()
            return lambda B, C: C.index_fill_(0, rows, 0.0)
        rowptr, col, val, rmap = A.rowptr.data_ptr(), A.col.data_ptr(), _ptr(A.val), _ptr(A.row_map)
        if A.tasks is None and A.row_map is None and A.core is None and A.strip is None and A.dense3 is None:
            nrows = A.nrows
            def simple(B, C):
                check(lib.pgcn_spmm_csr_f32(rowptr, col, val, nrows, B.data_ptr(), ldb, C.data_ptr(), ldc, f,
                                            flags, stream()), "pgcn_spmm_csr_f32")
            return simple
        need = A.nslots_total * f
        if need and (A.ws is None or A.ws.numel() < need):
            if A.ws is not None:
                A.retired_ws.append(A.ws)       # a HIP graph captured earlier (or a copy of this matrix made with dataclasses.replace)
            A.ws = torch.empty(need, dtype=torch.float32, device=self.device)   # still launches into the old one: it stays allocated
            A.launch_cache.clear()              # other bindings hold the old work-space pointer
        ws, ws_n = _ptr(A.ws), (0 if A.ws is None else A.ws.numel())
        tasks, ntasks, seg, nslices = _ptr(A.tasks), A.ntasks, A.seg, A.nslices
        nslots = A.nslots
        if A.core is None and A.strip is None and A.dense3 is None:
            fix, nfix = _ptr(A.fix), A.nfix
            def planned(B, C):
                check(lib.pgcn_spmm_csr_plan_f32(rowptr, col, val, tasks, ntasks, seg, nslices, fix, nfix, rmap,
                                                 B.data_ptr(), ldb, C.data_ptr(), ldc, f, ws, ws_n, nslots, flags,
                                                 stream()), "pgcn_spmm_csr_plan_f32")
            return planned
        # gather part (partial sums stay in the work-space) + LDS-tiled core + MFMA tiles + one combined fix-up
        co, st, d3 = A.core, A.strip, A.dense3
        if d3 is not None:
            need3 = int(lib.pgcn_dense_bf16x3_image_bytes(d3.npanels, f))
            if d3.image is None or d3.image.numel() < need3:
                if d3.image is not None:
                    d3.retired.append(d3.image)   # (scratch of one launch group: a stale binding stays correct, it only must not dangle)
                d3.image = torch.empty(need3, dtype=torch.uint8, device=self.device)
                A.launch_cache.clear()          # other bindings hold the old work-space pointer
            w3, n3, bi3, v3, pl3, np3, img3, imgb3 = (d3.work.data_ptr(), d3.npieces, d3.blk_img.data_ptr(), d3.vals3.data_ptr(),
                                                      d3.panel_list.data_ptr(), d3.npanels, d3.image.data_ptr(), d3.image.numel())
        if st is not None:
            sw, sn, srec, spairs = st.work.data_ptr(), st.npieces, st.rec.data_ptr(), st.pairs.data_ptr()
        if co is not None:
            cw, cn, ctp, ctb, cso, ccol, cval = (co.work.data_ptr(), co.npieces, co.tile_panel.data_ptr(),
                                                 co.tile_base.data_ptr(), co.seg_off.data_ptr(), co.ccol.data_ptr(),
                                                 co.cval.data_ptr())
        else:
            cw = cn = ctp = ctb = cso = ccol = cval = None
        ncols, nst = A.ncols, A.nslots_total
        fixp, nfa, slots = A.fix_all.data_ptr(), A.fix_all.shape[0], A.slot_ids.data_ptr()
        gflags, fflags = flags | _lib.SPMM_NO_FIXUP, flags & _lib.SPMM_ACCUMULATE

        # launch lanes (tuning.lanes): "strip/gather+dense3" = the strips on a second stream, the gather part and then the bf16 blocks
        # on the current one ("/" separates lanes, launched in the order written; the LAST lane is the current stream).  The producers
        # only write their own partial rows; the fix-up waits for every lane.  Fork and join are events: legal inside a HIP-graph capture.
        present = [n for n, on in (("gather", ntasks), ("strip", st is not None), ("dense3", d3 is not None)) if on]
        spec = "gather+strip+dense3" if (self.single_lane or not _T.lanes or A.nnz < _T.lanes_min_nnz) else _T.lanes
        lanes = [[x for x in lane.split("+") if x] for lane in spec.split("/")]
        if sorted(x for lane in lanes for x in lane) != ["dense3", "gather", "strip"]:
            raise _lib.PgcnError("tuning.lanes must name gather, strip and dense3 once each (lanes by /, order by +), got %r" % spec)
        lanes = [[x for x in lane if x in present] for lane in lanes]
        lanes = [lane for lane in lanes[:-1] if lane] + [lanes[-1]]
        while len(self.sides) < len(lanes) - 1:
            self.sides.append(torch.cuda.Stream(self.device))
        sides = self.sides
        dev, nside = self.device, len(lanes) - 1

        def launch(name, b, c, s):
            if name == "gather":
                check(lib.pgcn_spmm_csr_plan_f32(rowptr, col, val, tasks, ntasks, seg, nslices, None, 0, rmap, b, ldb,
                                                 c, ldc, f, ws, ws_n, nslots, gflags, s), "pgcn_spmm_csr_plan_f32")
            elif name == "strip":
                check(lib.pgcn_spmm_strip_f32(sw, sn, srec, spairs, b, ldb, ncols, f, ws, ws_n, nst, s), "pgcn_spmm_strip_f32")
            else:
                check(lib.pgcn_spmm_dense_bf16x3_f32(w3, n3, bi3, v3, pl3, np3, b, ldb, ncols, f, img3, imgb3, ws, ws_n, nst, s),
                      "pgcn_spmm_dense_bf16x3_f32")

        def produce(B, C):
            b, c, s = B.data_ptr(), C.data_ptr(), stream()
            if nside:
                cur = torch.cuda.current_stream(dev)
                for i in range(nside):
                    sides[i].wait_stream(cur)
                    o = sides[i].cuda_stream
                    for name in lanes[i]:
                        launch(name, b, c, o)
            for name in lanes[-1]:
                launch(name, b, c, s)
            if co is not None:
                check(lib.pgcn_spmm_core_f32(cw, cn, ctp, ctb, cso, ccol, cval, b, ldb, ncols, f, ws, ws_n, nst, s),
                      "pgcn_spmm_core_f32")
            for i in range(nside):
                cur.wait_stream(sides[i])

        def fixup(C):
            check(lib.pgcn_spmm_fixup_f32(fixp, nfa, slots, rmap, ws, C.data_ptr(), ldc, f, fflags, stream()), "pgcn_spmm_fixup_f32")

        def hybrid(B, C):
            produce(B, C)
            fixup(C)
        return hybrid

    # -- GAT path (pgcn_gat.hip) ---------------------------------------------
    def prepare_gat(self, csr: HostCSR, rows_wave: torch.Tensor, rows_block: torch.Tensor, chunk: Optional[int] = None,
                    small_row: Optional[int] = None) -> DeviceCSR:
        """Pattern structure for the attention kernels + SpMM plan; the values come per layer and
        head through ``with_values``."""
        if csr.core is not None or csr.row_map is not None:
            raise _lib.PgcnError("GAT structures are plain (sliced) CSR blocks")
        d = self.prepare(csr, pattern_only=True, chunk=chunk, small_row=small_row)        # (the attention kernels walk (row, slice) pieces)
        d.rows_wave = rows_wave.to(self.device, torch.int32).contiguous()
        d.rows_block = rows_block.to(self.device, torch.int32).contiguous()
        if csr.nslices == 8 and csr.slice_cnt is not None and csr.ngroups == 1:
            off = torch.zeros((csr.nrows, 9), dtype=torch.int32, device=csr.slice_cnt.device)
            off[:, 1:] = torch.cumsum(csr.slice_cnt.to(torch.int64), 1).to(torch.int32)
            d.slice_off = off.to(self.device).contiguous()
            ln = csr.rowptr[1:] - csr.rowptr[:-1]
            d.rows_all = torch.argsort(-ln, stable=True).to(torch.int32).to(self.device).contiguous()
        return d

    def with_values(self, A: DeviceCSR, plane: torch.Tensor) -> DeviceCSR:
        """The same structure with ``plane`` (fp32 [nnz], storage order) as its value array."""
        if plane.dtype is not torch.float32 or not plane.is_cuda or plane.dim() != 1 or plane.stride(0) != 1 \
                or plane.numel() < A.col.numel():
            raise _lib.PgcnError("value plane must be a contiguous fp32 CUDA vector of nnz entries")
        return dataclasses.replace(A, val=plane, launch_cache={}, ws=None, retired_ws=[])

    @staticmethod
    def _lists(A: DeviceCSR):
        return (_ptr(A.rows_wave), A.rows_wave.numel(), _ptr(A.rows_block), A.rows_block.numel())

    def _check_rows(self, t: torch.Tensor, rows: int, cols: int, what: str):
        if not (t.is_cuda and t.dtype is torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] >= rows
                and t.shape[1] == cols):
            raise _lib.PgcnError("%s must be an fp32 CUDA matrix with >= %d rows and %d unit-stride columns"
                                 % (what, rows, cols))

    def spmm_heads(self, A: DeviceCSR, alpha: torch.Tensor, B: torch.Tensor, C: torch.Tensor, heads: int, d: int,
                   accumulate: bool = False) -> bool:
        """C[:, k*d:(k+1)*d] (+)= A_{alpha_k} . B[:, k*d:(k+1)*d] for all heads in one launch
        (pgcn_spmm_heads_f32).  Returns False when the shape is not covered (the caller then runs one SpMM
        per head)."""
        F = heads * d
        nnz = A.col.numel()
        if not (alpha.is_cuda and alpha.dtype is torch.float32 and alpha.dim() == 2 and alpha.shape[0] == heads
                and alpha.stride(1) == 1 and (nnz == 0 or alpha.shape[1] >= nnz)):
            raise _lib.PgcnError("alpha must be [heads, nnz] fp32 CUDA planes")
        self._check_dense(B, A.ncols, "B")
        self._check_dense(C, A.nrows, "C")
        if B.shape[1] < F or C.shape[1] < F or A.row_map is not None:
            raise _lib.PgcnError("B / C narrower than heads * d, or a compact-row structure")
        need = A.nslots * F
        if need and (A.ws is None or A.ws.numel() < need):
            A.ws = torch.empty(need, dtype=torch.float32, device=self.device)
            A.launch_cache.clear()
        flags = _lib.SPMM_ACCUMULATE if accumulate else 0
        rc = self.lib.pgcn_spmm_heads_f32(
            A.rowptr.data_ptr(), A.col.data_ptr(), alpha.data_ptr(), alpha.stride(0), heads, d, A.nrows, _ptr(A.tasks),
            A.ntasks, A.seg, A.nslices, _ptr(A.fix), A.nfix, B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0),
            _ptr(A.ws), 0 if A.ws is None else A.ws.numel(), A.nslots, flags, self._stream())
        if rc == _lib.PGCN_EUNSUPPORTED:
            return False
        _lib.check(rc, "pgcn_spmm_heads_f32")
        return True

    def spmm_heads_recompute(self, AT: DeviceCSR, rowstat: torch.Tensor, s2: torch.Tensor, slope: float, mode: int,
                             B: torch.Tensor, C: torch.Tensor, heads: int, d: int, accumulate: bool = False) -> bool:
        """C[:, k*d:(k+1)*d] (+)= A_alpha_k^T . B[:, k*d:(k+1)*d] on the TRANSPOSED structure ``AT`` with the weights
        recomputed per entry from the softmax's row statistics (pgcn_spmm_heads_recompute_f32): no alpha^T planes.
        rowstat: [AT.ncols, heads, 4]; s2: [>= AT.nrows, heads] (s2 of the rows of AT).  False = shape not covered."""
        F = heads * d
        self._check_rows(s2, AT.nrows, heads, "s2")
        if not (rowstat.is_cuda and rowstat.dtype is torch.float32 and rowstat.is_contiguous()
                and rowstat.numel() == AT.ncols * heads * 4):
            raise _lib.PgcnError("rowstat must be a contiguous fp32 [ncols, heads, 4] CUDA tensor")
        self._check_dense(B, AT.ncols, "B")
        self._check_dense(C, AT.nrows, "C")
        if B.shape[1] < F or C.shape[1] < F or AT.row_map is not None:
            raise _lib.PgcnError("B / C narrower than heads * d, or a compact-row structure")
        need = AT.nslots * F
        if need and (AT.ws is None or AT.ws.numel() < need):
            AT.ws = torch.empty(need, dtype=torch.float32, device=self.device)
            AT.launch_cache.clear()
        flags = _lib.SPMM_ACCUMULATE if accumulate else 0
        rc = self.lib.pgcn_spmm_heads_recompute_f32(
            AT.rowptr.data_ptr(), AT.col.data_ptr(), rowstat.data_ptr(), s2.data_ptr(), s2.stride(0), slope, mode, heads, d,
            AT.nrows, _ptr(AT.tasks), AT.ntasks, AT.seg, AT.nslices, _ptr(AT.fix), AT.nfix, B.data_ptr(), B.stride(0),
            C.data_ptr(), C.stride(0), _ptr(AT.ws), 0 if AT.ws is None else AT.ws.numel(), AT.nslots, flags, self._stream())
        if rc == _lib.PGCN_EUNSUPPORTED:
            return False
        _lib.check(rc, "pgcn_spmm_heads_recompute_f32")
        return True

    def spmm_heads_forward2(self, A: DeviceCSR, rowstat: torch.Tensor, s2: torch.Tensor, slope: float, mode: int,
                            B: torch.Tensor, C: torch.Tensor, C2: torch.Tensor, heads: int, d: int,
                            accumulate: bool = False) -> bool:
        """The forward product with the weights recomputed from the row statistics (no alpha planes) and a second
        accumulator (pgcn_spmm_heads_forward2_f32): C[:, :F] (+)= A_alpha . B; C2[:, :F] (+)= V = sum_j c_ij B_j and
        C2[:, F:F+heads] (+)= sum_j c_ij with c = alpha x LeakyReLU' (standard) or alpha + beta (reference), from
        which the backward pass gets ds1 = <dOut, V> - t C without a per-entry gradient.  rowstat: [A.nrows, heads, 4];
        s2: [A.ncols, heads].  False = shape not covered (d must be 32, 64, 128 or 256)."""
        F = heads * d
        hl = d // 4
        if d % 4 or hl < 8 or hl & (hl - 1) or F > 256 or heads > 8:
            return False
        pw2 = F + (heads + 3) // 4 * 4
        self._check_rows(s2, A.ncols, heads, "s2")
        if not (rowstat.is_cuda and rowstat.dtype is torch.float32 and rowstat.is_contiguous()
                and rowstat.numel() == A.nrows * heads * 4):
            raise _lib.PgcnError("rowstat must be a contiguous fp32 [nrows, heads, 4] CUDA tensor")
        self._check_dense(B, A.ncols, "B")
        self._check_dense(C, A.nrows, "C")
        self._check_dense(C2, A.nrows, "C2")
        if B.shape[1] < F or C.shape[1] < F or C2.shape[1] < pw2 or A.row_map is not None:
            raise _lib.PgcnError("B / C narrower than heads * d, C2 narrower than heads * d + heads (rounded up to 4), "
                                 "or a compact-row structure")
        if A.col.numel() == 0:               # nothing stored (every entry of the pattern sits in blocks, r06): the outputs are defined all the same
            if not accumulate:
                C[:A.nrows, :F].zero_()
                C2[:A.nrows, :pw2].zero_()
            return True
        need = A.nslots * (F + pw2)
        if need and (A.ws is None or A.ws.numel() < need):
            A.ws = torch.empty(need, dtype=torch.float32, device=self.device)
            A.launch_cache.clear()
        flags = _lib.SPMM_ACCUMULATE if accumulate else 0
        rc = self.lib.pgcn_spmm_heads_forward2_f32(
            A.rowptr.data_ptr(), A.col.data_ptr(), rowstat.data_ptr(), s2.data_ptr(), s2.stride(0), slope, mode, heads, d,
            A.nrows, _ptr(A.tasks), A.ntasks, A.seg, A.nslices, _ptr(A.fix), A.nfix, B.data_ptr(), B.stride(0),
            C.data_ptr(), C.stride(0), C2.data_ptr(), C2.stride(0), _ptr(A.ws), 0 if A.ws is None else A.ws.numel(),
            A.nslots, flags, self._stream())
        if rc == _lib.PGCN_EUNSUPPORTED:
            return False
        _lib.check(rc, "pgcn_spmm_heads_forward2_f32")
        return True

    def spmm_heads_grad(self, AT: DeviceCSR, rowstat: torch.Tensor, s2: torch.Tensor, slope: float, mode: int,
                        B: torch.Tensor, Z: torch.Tensor, t: torch.Tensor, C: torch.Tensor, de: Optional[torch.Tensor],
                        heads: int, d: int, accumulate: bool = False) -> bool:
        """The transposed product of ``spmm_heads_recompute`` AND the edge gradient of the same (i, j) pairs in one
        gather pass (pgcn_spmm_heads_grad_f32): C[:, :F] (+)= A_alpha^T . B, de[q] = the edge gradient of entry q of
        ``AT`` (entry-major [nnz, heads]), C[:, F:F+heads] (+)= its row sums (ds2).  Z: rows of AT ([>= AT.nrows, >= F]),
        t: [AT.ncols, heads].  de = None: the per-entry gradient is not kept (ds1 then comes from the forward pass's
        second accumulator, ``spmm_heads_forward2``).  False = shape not covered (d must be 32, 64, 128 or 256)."""
        F = heads * d
        hl = d // 4
        if d % 4 or hl < 8 or hl & (hl - 1) or F > 256 or heads > 8:
            return False                   # (the C entry point answers PGCN_EUNSUPPORTED for these too)
        pw = F + (heads + 3) // 4 * 4
        nnz = AT.col.numel()
        self._check_rows(s2, AT.nrows, heads, "s2")
        self._check_rows(t, AT.ncols, heads, "t")
        if not (rowstat.is_cuda and rowstat.dtype is torch.float32 and rowstat.is_contiguous()
                and rowstat.numel() == AT.ncols * heads * 4):
            raise _lib.PgcnError("rowstat must be a contiguous fp32 [ncols, heads, 4] CUDA tensor")
        self._check_dense(B, AT.ncols, "B")
        self._check_dense(Z, AT.nrows, "Z")
        self._check_dense(C, AT.nrows, "C")
        if B.shape[1] < F or Z.shape[1] < F or C.shape[1] < pw or AT.row_map is not None or t.stride(0) != heads:
            raise _lib.PgcnError("B / Z narrower than heads * d, C narrower than heads * d + heads (rounded up to 4), "
                                 "t not contiguous, or a compact-row structure")
        if de is not None and not (de.is_cuda and de.dtype is torch.float32 and de.is_contiguous()
                                   and de.numel() >= nnz * heads):
            raise _lib.PgcnError("de must be a contiguous fp32 CUDA tensor of nnz * heads elements")
        if nnz == 0:
            if not accumulate:
                C[:AT.nrows, :pw].zero_()
            return True
        need = AT.nslots * pw
        if need and (AT.ws is None or AT.ws.numel() < need):
            AT.ws = torch.empty(need, dtype=torch.float32, device=self.device)
            AT.launch_cache.clear()
        flags = _lib.SPMM_ACCUMULATE if accumulate else 0
        rc = self.lib.pgcn_spmm_heads_grad_f32(
            AT.rowptr.data_ptr(), AT.col.data_ptr(), rowstat.data_ptr(), s2.data_ptr(), s2.stride(0), slope, mode, heads, d,
            AT.nrows, _ptr(AT.tasks), AT.ntasks, AT.seg, AT.nslices, _ptr(AT.fix), AT.nfix, B.data_ptr(), B.stride(0),
            Z.data_ptr(), Z.stride(0), t.data_ptr(), C.data_ptr(), C.stride(0), _ptr(de), _ptr(AT.ws),
            0 if AT.ws is None else AT.ws.numel(), AT.nslots, flags, self._stream())
        if rc == _lib.PGCN_EUNSUPPORTED:
            return False
        _lib.check(rc, "pgcn_spmm_heads_grad_f32")
        return True

    # -- the dense blocks of an attention pattern (pgcn_gat_blocks.hip, r06) ------------------------------------------
    @staticmethod
    def gat_block_bits(h3) -> torch.Tensor:
        """The pattern of ``HostDense3`` blocks as bits: [nblocks, 8 waves, 64 lanes, 4 words] int32 -- byte u = 2 ks + rb of a lane's
        16 bytes, bit 4 h + e, the A-operand order of ``partition.dense3_index`` (include/pgcn_hip.h, pgcn_gat_blocks_forward_f32)."""
        nb = h3.vals3.shape[0]
        nz = (h3.vals3 != 0).view(nb, 8, 16, 2, 64, 4)                                  # [block][w][unit][h][lane][e]
        wgt = (1 << (4 * torch.arange(2, device=nz.device).view(2, 1, 1) + torch.arange(4, device=nz.device).view(1, 1, 4))).to(torch.int64)
        byte = (nz.to(torch.int64) * wgt.view(1, 1, 1, 2, 1, 4)).sum((3, 5))            # [block][w][unit][lane]
        byte = byte.permute(0, 1, 3, 2).reshape(nb, 8, 64, 4, 4)                        # [block][w][lane][word][byte of the word]
        word = (byte << (8 * torch.arange(4, device=nz.device).view(1, 1, 1, 1, 4))).sum(-1)
        word = torch.where(word >= (1 << 31), word - (1 << 32), word)
        return word.to(torch.int32).contiguous()

    def prepare_gat_blocks(self, h3) -> DeviceGatBlocks:
        """Upload the blocks split off an attention pattern (``partition.split_dense3`` on the pattern's coordinates)."""
        dev = self.device
        work = h3.work.to(torch.int32)
        w64 = work.cpu().to(torch.int64)
        p0, pr = h3.piece_row0.cpu().to(torch.int64), h3.piece_rows.cpu().to(torch.int64)
        start = torch.cumsum(pr, 0) - pr
        rit = torch.arange(int(pr.sum()), dtype=torch.int64) - torch.repeat_interleave(start, pr)
        rows = torch.repeat_interleave(p0, pr) + rit
        slots = torch.repeat_interleave(w64[:, 3], pr) + rit
        seq = torch.repeat_interleave(torch.arange(w64.shape[0], dtype=torch.int64), pr)
        ok = rows < h3.nrows
        rows, slots, seq = rows[ok], slots[ok], seq[ok]
        order = torch.argsort(rows * (1 << 32) + seq)
        rows, slots = rows[order], slots[order]
        urows, counts = torch.unique_consecutive(rows, return_counts=True)
        begin = torch.cumsum(counts, 0) - counts
        fix = torch.stack([urows, begin, counts, torch.zeros_like(urows)], 1).to(torch.int32)
        return DeviceGatBlocks(h3.nrows, h3.ncols, work.to(dev).contiguous(), h3.piece_row0.to(dev, torch.int32).contiguous(),
                               h3.blk_img.to(dev).contiguous(), self.gat_block_bits(h3).to(dev), h3.panel_list.to(dev).contiguous(),
                               fix.to(dev).contiguous(), slots.to(torch.int32).to(dev).contiguous(), h3.npieces,
                               int(h3.panel_list.numel()), h3.nslots, h3.nnz)

    def _gat_blocks_ws(self, G: DeviceGatBlocks, F: int, per_slot: int):
        need_img = int(self.lib.pgcn_dense_bf16x3_image_bytes(G.npanels, F))
        if G.image is None or G.image.numel() < need_img:
            G.image = torch.empty(need_img, dtype=torch.uint8, device=self.device)
        need = G.nslots * per_slot
        if G.ws is None or G.ws.numel() < need:
            G.ws = torch.empty(need, dtype=torch.float32, device=self.device)

    def gat_blocks_forward(self, G: DeviceGatBlocks, rowstat: torch.Tensor, s2: torch.Tensor, slope: float, B: torch.Tensor,
                           C: torch.Tensor, C2: torch.Tensor, heads: int, d: int) -> bool:
        """C[:, :F] += the blocks' part of A_alpha . B, C2 += their part of V | C (the outputs of ``spmm_heads_forward2``, which must have
        run over the remaining entries before: this call accumulates).  False = shape not covered (d = 64, standard mode only)."""
        F = heads * d
        if d != 64 or F > 256:
            return False
        pw2 = F + (heads + 3) // 4 * 4
        self._check_rows(s2, G.ncols, heads, "s2")
        if not (rowstat.is_cuda and rowstat.dtype is torch.float32 and rowstat.is_contiguous() and rowstat.numel() == G.nrows * heads * 4):
            raise _lib.PgcnError("rowstat must be a contiguous fp32 [nrows, heads, 4] CUDA tensor")
        self._check_dense(B, G.ncols, "B")
        self._check_dense(C, G.nrows, "C")
        self._check_dense(C2, G.nrows, "C2")
        if B.shape[1] < F or C.shape[1] < F or C2.shape[1] < pw2:
            raise _lib.PgcnError("B / C narrower than heads * d or C2 narrower than heads * d + heads (rounded up to 4)")
        self._gat_blocks_ws(G, F, F + pw2)
        st = self._stream()
        rc = self.lib.pgcn_gat_blocks_forward_f32(
            G.work.data_ptr(), G.npieces, G.work_row0.data_ptr(), G.blk_img.data_ptr(), G.bits.data_ptr(), G.panel_list.data_ptr(),
            G.npanels, rowstat.data_ptr(), s2.data_ptr(), s2.stride(0), slope, heads, d, G.nrows, G.ncols, B.data_ptr(), B.stride(0),
            G.image.data_ptr(), G.image.numel(), G.ws.data_ptr(), G.ws.numel(), G.nslots, st)
        if rc == _lib.PGCN_EUNSUPPORTED:
            return False
        _lib.check(rc, "pgcn_gat_blocks_forward_f32")
        nfix = G.fix.shape[0]
        _lib.check(self.lib.pgcn_spmm_fixup_f32(G.fix.data_ptr(), nfix, G.slot_ids.data_ptr(), None, G.ws.data_ptr(), C.data_ptr(),
                                                C.stride(0), F, _lib.SPMM_ACCUMULATE, st), "pgcn_spmm_fixup_f32")
        _lib.check(self.lib.pgcn_spmm_fixup_f32(G.fix.data_ptr(), nfix, G.slot_ids.data_ptr(), None, G.ws.data_ptr() + 4 * G.nslots * F,
                                                C2.data_ptr(), C2.stride(0), pw2, _lib.SPMM_ACCUMULATE, st), "pgcn_spmm_fixup_f32")
        return True

    def gat_blocks_backward(self, G: DeviceGatBlocks, rowstat: torch.Tensor, s2: torch.Tensor, slope: float, B: torch.Tensor,
                            Z: torch.Tensor, t: torch.Tensor, C: torch.Tensor, heads: int, d: int) -> bool:
        """C[:, :F] += the blocks' part of A_alpha^T . B and C[:, F:F+heads] += their part of ds2 (the outputs of ``spmm_heads_grad`` with
        de = None, which must have run over the remaining entries before).  ``G``: blocks of the TRANSPOSED pattern; rowstat / t of its
        columns, s2 / Z of its rows.  False = shape not covered."""
        F = heads * d
        if d != 64 or F > 256:
            return False
        pw = F + (heads + 3) // 4 * 4
        self._check_rows(s2, G.nrows, heads, "s2")
        self._check_rows(t, G.ncols, heads, "t")
        if not (rowstat.is_cuda and rowstat.dtype is torch.float32 and rowstat.is_contiguous() and rowstat.numel() == G.ncols * heads * 4):
            raise _lib.PgcnError("rowstat must be a contiguous fp32 [ncols, heads, 4] CUDA tensor")
        self._check_dense(B, G.ncols, "B")
        self._check_dense(Z, G.nrows, "Z")
        self._check_dense(C, G.nrows, "C")
        if B.shape[1] < F or Z.shape[1] < F or C.shape[1] < pw or t.stride(0) != heads:
            raise _lib.PgcnError("B / Z narrower than heads * d, C narrower than heads * d + heads (rounded up to 4) or t not contiguous")
        self._gat_blocks_ws(G, F, pw)
        st = self._stream()
        rc = self.lib.pgcn_gat_blocks_backward_f32(
            G.work.data_ptr(), G.npieces, G.work_row0.data_ptr(), G.blk_img.data_ptr(), G.bits.data_ptr(), G.panel_list.data_ptr(),
            G.npanels, rowstat.data_ptr(), s2.data_ptr(), s2.stride(0), t.data_ptr(), Z.data_ptr(), Z.stride(0), slope, heads, d,
            G.nrows, G.ncols, B.data_ptr(), B.stride(0), G.image.data_ptr(), G.image.numel(), G.ws.data_ptr(), G.ws.numel(), G.nslots, st)
        if rc == _lib.PGCN_EUNSUPPORTED:
            return False
        _lib.check(rc, "pgcn_gat_blocks_backward_f32")
        _lib.check(self.lib.pgcn_spmm_fixup_f32(G.fix.data_ptr(), G.fix.shape[0], G.slot_ids.data_ptr(), None, G.ws.data_ptr(), C.data_ptr(),
                                                C.stride(0), pw, _lib.SPMM_ACCUMULATE, st), "pgcn_spmm_fixup_f32")
        return True

    def gat_edge_softmax(self, A: DeviceCSR, s1, s2, heads: int, slope: float, mode: int, n_global: int,
                         alpha: Optional[torch.Tensor], beta: torch.Tensor, rowstat: Optional[torch.Tensor] = None) -> None:
        """alpha = None: the per-row statistics only (``rowstat`` required) -- the products recompute the weights."""
        self._check_rows(s1, A.nrows, heads, "s1")
        self._check_rows(s2, A.ncols, heads, "s2")
        self._check_rows(beta, A.nrows, heads, "beta")
        nnz = A.col.numel()
        if alpha is None:
            if rowstat is None:
                raise _lib.PgcnError("alpha = None needs rowstat")
        else:
            self._check_rows(alpha, heads, alpha.shape[1], "alpha")
            if (nnz and alpha.shape[1] != nnz) or alpha.stride(0) != alpha.shape[1]:
                raise _lib.PgcnError("alpha must be [heads, nnz], contiguous")
        if beta.stride(0) != heads:
            raise _lib.PgcnError("beta must be [nrows, heads], contiguous")
        if rowstat is not None and not (rowstat.is_cuda and rowstat.dtype is torch.float32 and rowstat.is_contiguous()
                                        and rowstat.numel() == A.nrows * heads * 4):
            raise _lib.PgcnError("rowstat must be a contiguous fp32 [nrows, heads, 4] CUDA tensor")
        lists = self._lists(A)
        _lib.check(self.lib.pgcn_gat_edge_softmax_f32(
            A.rowptr.data_ptr(), A.col.data_ptr(), A.nrows, nnz, *lists, s1.data_ptr(),
            s1.stride(0), s2.data_ptr(), s2.stride(0), heads, slope, mode, n_global, _ptr(alpha), beta.data_ptr(),
            _ptr(rowstat), self._stream()), "pgcn_gat_edge_softmax_f32")

    def gat_edge_weights_t(self, AT: DeviceCSR, s2, rowstat: torch.Tensor, heads: int, slope: float, mode: int,
                           alpha_t: torch.Tensor) -> None:
        """alpha planes in the storage order of the transposed structure ``AT`` (rows = columns of A)."""
        self._check_rows(s2, AT.nrows, heads, "s2")
        nnz = AT.col.numel()
        if alpha_t.dim() != 2 or alpha_t.shape[0] != heads or not alpha_t.is_contiguous() \
                or (nnz and alpha_t.shape[1] != nnz):
            raise _lib.PgcnError("alpha_t must be [heads, nnz] contiguous")
        if not (rowstat.is_cuda and rowstat.dtype is torch.float32 and rowstat.is_contiguous()
                and rowstat.numel() == AT.ncols * heads * 4):
            raise _lib.PgcnError("rowstat must be a contiguous fp32 [ncols, heads, 4] CUDA tensor")
        _lib.check(self.lib.pgcn_gat_edge_weights_t_f32(
            AT.rowptr.data_ptr(), AT.col.data_ptr(), AT.nrows, nnz, *self._lists(AT), s2.data_ptr(), s2.stride(0),
            rowstat.data_ptr(), heads, slope, mode, alpha_t.data_ptr(), self._stream()), "pgcn_gat_edge_weights_t_f32")

    @staticmethod
    def _bad_de(de, alpha, heads: int, nnz: int) -> bool:
        """alpha: [heads, nnz] head-major planes; de: [nnz, heads] ENTRY-major (include/pgcn_hip.h)."""
        return (alpha.dim() != 2 or de.dim() != 2 or alpha.shape[0] != heads or de.shape[1] != heads
                or de.shape[0] != alpha.shape[1] or not de.is_contiguous() or not alpha.is_contiguous()
                or bool(nnz and alpha.shape[1] != nnz))

    def gat_edge_grad(self, A: DeviceCSR, s1, s2, alpha, beta, Z, dOut, t, heads: int, d: int, slope: float,
                      mode: int, de: torch.Tensor, ds1: torch.Tensor) -> None:
        self._check_rows(s1, A.nrows, heads, "s1")
        self._check_rows(s2, A.ncols, heads, "s2")
        self._check_rows(t, A.nrows, heads, "t")
        self._check_rows(ds1, A.nrows, heads, "ds1")
        self._check_rows(dOut, A.nrows, heads * d, "dOut")
        if not (Z.is_cuda and Z.dtype is torch.float32 and Z.dim() == 2 and Z.stride(1) == 1 and Z.shape[0] >= A.ncols
                and Z.shape[1] >= heads * d):
            raise _lib.PgcnError("Z must hold ncols rows of at least heads*d fp32 columns")
        nnz = A.col.numel()
        if self._bad_de(de, alpha, heads, nnz) \
                or t.stride(0) != heads or ds1.stride(0) != heads or beta.stride(0) != heads:
            raise _lib.PgcnError("alpha must be [heads, nnz], de [nnz, heads], both contiguous; t, ds1, beta [nrows, heads] contiguous")
        _lib.check(self.lib.pgcn_gat_edge_grad_f32(
            A.rowptr.data_ptr(), A.col.data_ptr(), A.nrows, nnz, *self._lists(A), s1.data_ptr(),
            s1.stride(0), s2.data_ptr(), s2.stride(0), alpha.data_ptr(), beta.data_ptr(), Z.data_ptr(), Z.stride(0),
            dOut.data_ptr(), dOut.stride(0), t.data_ptr(), heads, d, slope, mode, de.data_ptr(), ds1.data_ptr(),
            self._stream()), "pgcn_gat_edge_grad_f32")

    def gat_edge_grad_sliced(self, A: DeviceCSR, s1, s2, alpha, beta, Z, dOut, t, heads: int, d: int, slope: float,
                             mode: int, de: torch.Tensor, ds1_slices: torch.Tensor) -> bool:
        """XCD-sliced variant (structure stored in 8 col % 8 slices).  Returns False when the shape is
        not covered (the caller then uses gat_edge_grad); ds1_slices: [nrows, 8, heads]."""
        F = heads * d
        if A.slice_off is None or d % 4 or F > 256 or (d // 4) & (d // 4 - 1) or Z.stride(0) % 4 or dOut.stride(0) % 4 \
                or Z.data_ptr() % 16 or dOut.data_ptr() % 16:
            return False
        self._check_rows(s1, A.nrows, heads, "s1")
        self._check_rows(s2, A.ncols, heads, "s2")
        self._check_rows(t, A.nrows, heads, "t")
        self._check_rows(dOut, A.nrows, F, "dOut")
        nnz = A.col.numel()
        if self._bad_de(de, alpha, heads, nnz) \
                or t.stride(0) != heads or beta.stride(0) != heads or not ds1_slices.is_contiguous() \
                or tuple(ds1_slices.shape) != (A.nrows, 8, heads) or Z.shape[0] < A.ncols or Z.shape[1] < F:
            raise _lib.PgcnError("bad operand shapes for the sliced edge gradient")
        _lib.check(self.lib.pgcn_gat_edge_grad_sliced_f32(
            A.rowptr.data_ptr(), A.col.data_ptr(), A.slice_off.data_ptr(), A.nrows, nnz, A.rows_all.data_ptr(), A.nrows,
            s1.data_ptr(), s1.stride(0), s2.data_ptr(), s2.stride(0), alpha.data_ptr(), beta.data_ptr(), Z.data_ptr(),
            Z.stride(0), dOut.data_ptr(), dOut.stride(0), t.data_ptr(), heads, d, slope, mode, de.data_ptr(),
            ds1_slices.data_ptr(), self._stream()), "pgcn_gat_edge_grad_sliced_f32")
        return True

    def gat_edge_grad_tasks(self, A: DeviceCSR, s1, s2, alpha, beta, Z, dOut, t, heads: int, d: int, slope: float,
                            mode: int, de: torch.Tensor, ds1: torch.Tensor) -> bool:
        """The edge gradient over the tasks of A's SpMM plan (balanced like the SpMM: pgcn_gat_edge_grad_tasks_f32).
        Returns False when the shape is not covered; ds1: [nrows, heads] contiguous, complete on return."""
        F = heads * d
        if d % 4 or F > 256 or (d // 4) & (d // 4 - 1) or Z.stride(0) % 4 or dOut.stride(0) % 4 \
                or Z.data_ptr() % 16 or dOut.data_ptr() % 16 or A.row_map is not None:
            return False
        self._check_rows(s1, A.nrows, heads, "s1")
        self._check_rows(s2, A.ncols, heads, "s2")
        self._check_rows(t, A.nrows, heads, "t")
        self._check_rows(ds1, A.nrows, heads, "ds1")
        self._check_rows(dOut, A.nrows, F, "dOut")
        nnz = A.col.numel()
        if self._bad_de(de, alpha, heads, nnz) \
                or t.stride(0) != heads or ds1.stride(0) != heads or beta.stride(0) != heads or Z.shape[0] < A.ncols \
                or Z.shape[1] < F:
            raise _lib.PgcnError("bad operand shapes for the task-based edge gradient")
        need = A.nslots * max(F, heads)
        if need and (A.ws is None or A.ws.numel() < need):
            A.ws = torch.empty(need, dtype=torch.float32, device=self.device)
            A.launch_cache.clear()
        rc = self.lib.pgcn_gat_edge_grad_tasks_f32(
            A.rowptr.data_ptr(), A.col.data_ptr(), A.nrows, nnz, _ptr(A.tasks), A.ntasks, A.seg, A.nslices, _ptr(A.fix),
            A.nfix, s1.data_ptr(), s1.stride(0), s2.data_ptr(), s2.stride(0), alpha.data_ptr(), beta.data_ptr(), Z.data_ptr(),
            Z.stride(0), dOut.data_ptr(), dOut.stride(0), t.data_ptr(), heads, d, slope, mode, de.data_ptr(), ds1.data_ptr(),
            _ptr(A.ws), 0 if A.ws is None else A.ws.numel(), A.nslots, self._stream())
        if rc == _lib.PGCN_EUNSUPPORTED:
            return False
        _lib.check(rc, "pgcn_gat_edge_grad_tasks_f32")
        return True

    def gat_row_dots(self, dOut: torch.Tensor, out: torch.Tensor, VC: Optional[torch.Tensor], heads: int, d: int):
        """(t, ds1) of the GAT backward in one pass (pgcn_gat_row_dots_f32); ds1 is None without VC; None when the kernel does not
        take the shapes (the caller keeps its tensor expressions)."""
        n = dOut.shape[0]
        t = torch.empty((n, heads), dtype=torch.float32, device=self.device)
        ds1 = torch.empty((n, heads), dtype=torch.float32, device=self.device) if VC is not None else None
        rc = self.lib.pgcn_gat_row_dots_f32(dOut.data_ptr(), dOut.stride(0), out.data_ptr(), out.stride(0), _ptr(VC),
                                            VC.stride(0) if VC is not None else 0, n, heads, d, t.data_ptr(), _ptr(ds1), self._stream())
        if rc == _lib.PGCN_EUNSUPPORTED:
            return None
        _lib.check(rc, "pgcn_gat_row_dots_f32")
        return t, ds1

    def csr_row_sums(self, A: DeviceCSR, perm: Optional[torch.Tensor], src: torch.Tensor, planes: int,
                     out: torch.Tensor) -> None:
        self._check_rows(out, A.nrows, planes, "out")
        nnz = A.col.numel()
        if src.dim() != 2 or src.shape[1] != planes or not src.is_contiguous() or (nnz and src.shape[0] != nnz):
            raise _lib.PgcnError("src must be [nnz, planes] contiguous (entry-major, like de)")
        if perm is not None and (perm.dtype is not torch.int64 or perm.numel() < A.col.numel()):
            raise _lib.PgcnError("perm must be int64 [nnz]")
        _lib.check(self.lib.pgcn_csr_row_sums_f32(
            A.rowptr.data_ptr(), _ptr(perm), A.nrows, nnz, *self._lists(A), src.data_ptr(), planes,
            out.data_ptr(), out.stride(0), self._stream()), "pgcn_csr_row_sums_f32")

    def csr_permute(self, src: torch.Tensor, perm: torch.Tensor, dst: torch.Tensor) -> None:
        if src.shape != dst.shape or src.dim() != 2 or not src.is_contiguous() or not dst.is_contiguous() \
                or perm.dtype is not torch.int64:
            raise _lib.PgcnError("src/dst must be [planes, nnz] contiguous, perm int64")
        n = perm.numel()
        if n == 0:
            return
        if src.shape[1] != n:                         # planes are padded when nnz == 0 only
            raise _lib.PgcnError("perm length must equal the plane length")
        _lib.check(self.lib.pgcn_csr_permute_f32(src.data_ptr(), perm.data_ptr(), n, src.shape[0], dst.data_ptr(),
                                                 self._stream()), "pgcn_csr_permute_f32")

    def nll_rows(self, X: torch.Tensor, labels: torch.Tensor):
        """(loss_rows, lse_rows) of nll_loss(log_softmax(X), labels) per row, or None when the shape is not covered."""
        if X.dim() != 2 or X.shape[1] > 1024 or X.stride(1) != 1 or labels.dtype is not torch.int64 or not labels.is_contiguous():
            return None
        self._check_dense(X, labels.numel(), "X")
        if X.shape[0] != labels.numel():
            raise _lib.PgcnError("nll_rows: %d rows of logits for %d labels" % (X.shape[0], labels.numel()))
        n, f = labels.numel(), X.shape[1]
        loss = torch.empty(n, dtype=torch.float32, device=self.device)
        lse = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pgcn_nll_rows_f32(X.data_ptr(), X.stride(0), labels.data_ptr(), n, f, loss.data_ptr(),
                                              lse.data_ptr(), self._stream()), "pgcn_nll_rows_f32")
        return loss, lse

    def nll_rows_backward(self, X: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor, gscale: torch.Tensor,
                          scale: float) -> torch.Tensor:
        n, f = labels.numel(), X.shape[1]
        if X.shape[0] != n:
            raise _lib.PgcnError("nll_rows_backward: %d rows of logits for %d labels" % (X.shape[0], n))
        dX = torch.empty((n, f), dtype=torch.float32, device=self.device)
        g = gscale.reshape(1).to(torch.float32).contiguous()
        _lib.check(self.lib.pgcn_nll_rows_backward_f32(X.data_ptr(), X.stride(0), labels.data_ptr(), lse.data_ptr(),
                                                       g.data_ptr(), scale, n, f, dX.data_ptr(), dX.stride(0),
                                                       self._stream()), "pgcn_nll_rows_backward_f32")
        return dX

    def gather_rows(self, H: torch.Tensor, idx: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        n = idx.numel()
        if n == 0:
            return out
        self._check_dense(H, 0, "H")
        self._check_dense(out, n, "out")
        _lib.check(self.lib.pgcn_gather_rows_f32(H.data_ptr(), H.stride(0), idx.data_ptr(), n,
                                                 out.data_ptr(), out.stride(0), H.shape[1],
                                                 self._stream()), "pgcn_gather_rows_f32")
        return out

    def scatter_rows(self, H: torch.Tensor, idx: torch.Tensor, src: torch.Tensor, accumulate: bool) -> torch.Tensor:
        n = idx.numel()
        if n == 0:
            return H
        self._check_dense(H, 0, "H")
        self._check_dense(src, n, "src")
        _lib.check(self.lib.pgcn_scatter_rows_f32(H.data_ptr(), H.stride(0), idx.data_ptr(), n,
                                                  src.data_ptr(), src.stride(0), H.shape[1],
                                                  int(accumulate), self._stream()), "pgcn_scatter_rows_f32")
        return H

    def device_info(self):
        out = (ctypes.c_int64 * 4)()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.pgcn_device_info(idx, out), "pgcn_device_info")
        return {"cus": out[0], "wave": out[1], "gfx": out[2], "l2_bytes": out[3]}
