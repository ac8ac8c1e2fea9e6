"""TEST INFRASTRUCTURE: a numpy walk of the launch group's DATA STRUCTURES, exactly as include/pgcn_hip.h documents them.

The HIP kernels consume a host-built plan: gather tasks + fix records (pgcn_spmm_plan_host), strip records
(partition.build_strips), LDS-core tiles, bf16 blocks (partition.build_dense3) and the combined per-row slot lists
(kernels.HipKernels._attach_core).  Everything up to the upload is host code, and a wrong offset in it only shows on a
GPU.  ``HostPlanner`` runs that host code without a device (it borrows ``HipKernels.prepare`` / ``_attach_core``
unchanged -- they only need ``self.device``), ``run_plan`` then executes the plan the way the kernels are specified to:
every task, record, tile and fix record is decoded from the arrays the device would get, in float64, so that the only
admissible difference to ``A @ B`` is float64 round-off.  No product code path uses this module."""
import importlib

import numpy as np
import torch

PKG = "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd"


class HostPlanner:
    """The host half of ``HipKernels`` (plan building, slot lists), tensors left on the CPU."""

    def __init__(self, chunk=None, small_row=None, adaptive_chunk=None):
        k = importlib.import_module(PKG + ".kernels")
        self.device = torch.device("cpu")
        self.chunk = k.DEFAULT_CHUNK if chunk is None else chunk
        self.small_row = k.DEFAULT_SMALL_ROW if small_row is None else small_row
        self.adaptive_chunk = k._T.spmm_adaptive_chunk if adaptive_chunk is None else adaptive_chunk
        self._k = k

    def prepare(self, csr, pattern_only=False, chunk=None, small_row=None):
        return self._k.HipKernels.prepare(self, csr, pattern_only, chunk, small_row)

    def prepare_gat(self, csr, rows_wave, rows_block, chunk=None, small_row=None):
        return self._k.HipKernels.prepare_gat(self, csr, rows_wave, rows_block, chunk, small_row)

    def _attach_core(self, d, csr, fix_rem):
        return self._k.HipKernels._attach_core(self, d, csr, fix_rem)


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def run_plan(d, B, C0=None, accumulate=False, checks=True, slice_of=None):
    """C (+)= A . B by walking the device-side arrays of ``d`` (a DeviceCSR whose tensors live on the CPU).
    Returns (C, info); rows nobody wrote stay NaN when ``accumulate`` is False (the caller decides what they must be).
    ``slice_of(cols)``: the slice of a column when the block was built with range slices (default col % nslices)."""
    part = importlib.import_module(PKG + ".partition")
    B = np.asarray(B, np.float64)
    f = B.shape[1]
    row_map = _np(d.row_map)
    nout = (int(row_map.max()) + 1 if row_map is not None and row_map.size else d.nrows) if C0 is None else C0.shape[0]
    C = np.full((nout, f), np.nan) if C0 is None else np.array(C0, np.float64)
    if C0 is None and accumulate:
        raise ValueError("accumulate needs C0")
    col, val = _np(d.col).astype(np.int64), (np.ones(d.col.numel()) if d.val is None else _np(d.val).astype(np.float64))
    rowptr = _np(d.rowptr)
    ws = np.full((max(d.nslots_total, d.nslots, 1), f), np.nan)
    info = {"entries_gather": 0, "entries_strip": 0, "entries_core": 0, "entries_dense": 0}

    def out_row(r):
        return int(row_map[r]) if row_map is not None else r

    def emit(r, s):
        o = out_row(r)
        C[o] = (C[o] + s) if accumulate else s

    tiled = d.core is not None or d.strip is not None or getattr(d, "dense3", None) is not None
    # ---- gather part -----------------------------------------------------------------------------------------
    if d.tasks is None:                                  # one task per row, no plan (pgcn_spmm_csr_f32)
        for r in range(d.nrows):
            k0, k1 = int(rowptr[r]), int(rowptr[r + 1])
            emit(r, (val[k0:k1, None] * B[col[k0:k1]]).sum(0))
            info["entries_gather"] += k1 - k0
    else:
        tasks = _np(d.tasks).astype(np.int64)
        covered = np.zeros(col.shape[0], np.int32)
        seg = [int(x) for x in d.seg]
        if checks:
            assert seg[0] == 0 and seg[-1] == tasks.shape[0] and all(a <= b for a, b in zip(seg, seg[1:]))
        row_of = np.searchsorted(rowptr, np.arange(col.shape[0]), side="right") - 1
        for s in range(len(seg) - 1):
            for lo, hi, length, dst in tasks[seg[s]:seg[s + 1]]:
                k0 = int(lo & 0xffffffff) | (int(hi) << 32)
                sl = slice(k0, k0 + int(length))
                covered[sl] += 1
                if checks and length:
                    rows = row_of[sl]
                    assert rows[0] == rows[-1], "a task spans two rows"
                    if d.nslices > 1:       # a task of segment s stays inside slice s of its row -- or is a whole short row
                        whole = k0 == rowptr[rows[0]] and k0 + length == rowptr[rows[0] + 1]
                        sls = col[sl] % d.nslices if slice_of is None else slice_of(col[sl])
                        assert whole or (sls == s).all()
                acc = (val[sl, None] * B[col[sl]]).sum(0)
                if dst >= 0:
                    assert np.isnan(ws[dst]).all(), "two tasks share a partial-sum slot"
                    ws[dst] = acc
                else:
                    r = int(~dst)
                    if checks and length:
                        assert row_of[k0] == r
                    emit(r, acc)
        if checks:
            assert (covered == 1).all(), "%d stored entries are not covered exactly once" % int((covered != 1).sum())
        info["entries_gather"] = int(col.shape[0])
    # ---- strip records (pgcn_spmm_strip_f32) -----------------------------------------------------------------
    if d.strip is not None:
        TR, NG, RW, SB, PAD = part.STRIP_TR, part.STRIP_NG, part.STRIP_RW, part.STRIP_B, part.STRIP_PAD_OFF
        work, rec, pairs = _np(d.strip.work), _np(d.strip.rec), _np(d.strip.pairs)
        inrec = np.arange(TR * SB)
        local_row = ((inrec // SB) % RW) * NG + inrec // (RW * SB)
        seen = np.zeros(rec.shape[0], np.int32)
        for tr, kb, ke, slot0 in work:
            acc = np.zeros((TR, f))
            prev_panel, run_starts = None, []
            for k in range(kb, ke):
                panel, same = int(rec[k, 0]), int(rec[k, 1])
                seen[k] += 1
                if same:
                    assert prev_panel == panel and k > kb, "a record claims a panel nobody staged"
                else:
                    assert not (k > kb and prev_panel == panel), "a panel is staged twice in a row"
                    run_starts.append(k)
                prev_panel = panel
                base = part.strip_panel_base(panel, d.ncols)
                off, bits = pairs[k, :, 0].astype(np.int64), pairs[k, :, 1]
                real = off != PAD
                if checks:
                    assert (off[real] % 512 == 0).all() and (off[real] >= 0).all() and (off[real] < PAD).all()
                    assert (bits[~real] == 0).all(), "an unused pair slot must hold the value 0.0"
                cols = base + off[real] // 512
                vals = bits[real].view(np.float32).astype(np.float64)
                rows_g = tr * TR + local_row[real]
                if checks:
                    assert (cols >= 0).all() and (cols < d.ncols).all() and (rows_g < d.nrows).all()
                np.add.at(acc, local_row[real], vals[:, None] * B[cols])
                info["entries_strip"] += int(real.sum())
            if checks:                                     # the panel of the NEXT run rides on every run start
                for a, b in zip(run_starts, run_starts[1:] + [None]):
                    assert int(rec[a, 2]) == (-1 if b is None else int(rec[b, 0]))
            assert np.isnan(ws[slot0:slot0 + TR]).all(), "two pieces share partial-sum slots"
            ws[slot0:slot0 + TR] = acc
        if checks:
            assert (seen == 1).all(), "strip records not covered exactly once"
    # ---- LDS-core tiles (pgcn_spmm_core_f32) -----------------------------------------------------------------
    if d.core is not None:
        TR, TC, NG, RW = part.CORE_TR, part.CORE_TC, part.CORE_NG, part.CORE_RW
        co = d.core
        work, tp, tb, so = _np(co.work), _np(co.tile_panel), _np(co.tile_base), _np(co.seg_off).astype(np.int64)
        ccol, cval = _np(co.ccol).astype(np.int64), _np(co.cval).astype(np.float64)
        ordn = np.arange(TR)
        rit = (ordn % RW) * NG + ordn // RW
        for tr, kb, ke, slot0 in work:
            acc = np.zeros((TR, f))
            for t in range(kb, ke):
                base = int(tb[t])
                for q in range(TR):
                    a, b = base + so[t, q], base + so[t, q + 1]
                    if b > a:
                        cols = int(tp[t]) * TC + ccol[a:b]
                        assert (cols < d.ncols).all()
                        acc[rit[q]] += (cval[a:b, None] * B[cols]).sum(0)
                        info["entries_core"] += int(b - a)
            assert np.isnan(ws[slot0:slot0 + TR]).all()
            ws[slot0:slot0 + TR] = acc
    # ---- bf16 three-plane blocks (pgcn_spmm_dense_bf16x3_f32): 512 x 128, A-operand order of v_mfma_f32_32x32x16_bf16 ----
    if getattr(d, "dense3", None) is not None:
        BR = part.DENSE3_BR
        d3 = d.dense3
        work, bimg, plist, vals = _np(d3.work), _np(d3.blk_img), _np(d3.panel_list), _np(d3.vals3).astype(np.float64)
        assert (np.diff(plist) > 0).all() and bimg.max() < plist.size
        i, k = np.meshgrid(np.arange(BR), np.arange(128), indexing="ij")
        idx = part.dense3_index(i, k)
        for br, first, cnt, slot0 in work:
            acc = np.zeros((BR, f))
            for t in range(first, first + cnt):
                blk = vals[t][idx]                                         # [row in block, column in panel]
                c0 = int(plist[bimg[t]])                                   # first row of the panel (any origin since r06)
                w = min(128, d.ncols - c0)
                assert w > 0 and not blk[:, w:].any(), "values beyond the last column of the block"
                acc += blk[:, :w] @ B[c0:c0 + w]
                info["entries_dense"] += int((blk != 0).sum())
            assert np.isnan(ws[slot0:slot0 + BR]).all()
            ws[slot0:slot0 + BR] = acc
    # ---- fix-up: a row's partial sums in the fixed order of its slot list ------------------------------------
    if tiled:
        fix, slots = _np(d.fix_all).astype(np.int64), _np(d.slot_ids).astype(np.int64)
        used = np.zeros(ws.shape[0], np.int32)
        for r, begin, cnt, _ in fix:
            ids = slots[begin:begin + cnt]
            used[ids] += 1
            assert not np.isnan(ws[ids]).any(), "the fix-up reads a slot nobody wrote"
            emit(int(r), ws[ids].sum(0))
        if checks:                      # every written slot of an existing row is read exactly once
            written = ~np.isnan(ws).any(1)
            assert (used[~written] == 0).all() and (used <= 1).all()
        info["fix_rows"] = int(fix.shape[0])
    elif d.tasks is not None and d.fix is not None:
        for r, first, cnt, _ in _np(d.fix).astype(np.int64):
            assert not np.isnan(ws[first:first + cnt]).any()
            emit(int(r), ws[first:first + cnt].sum(0))
        info["fix_rows"] = int(d.nfix)
    return C, info
