"""Every tunable of the aggregation path in ONE place, read ONCE.

The defaults are the measured choices of rounds 1-3 (DESIGN.md section 4 gives the sweep behind each).  A run can
override any field through the single environment variable

    PGCN_TUNING="strip_pieces=256,strip_min_records=0,exchange_rounds=1"

(comma separated ``field=value``; booleans as 0/1) -- this is what the probe scripts under ``tools/`` use.  Nothing
else in the package reads a tuning value from the environment.  The remaining ``PGCN_*`` variables select a
transport or an ingest path, not a kernel shape: PGCN_EXCHANGE, PGCN_OVERLAP, PGCN_INGEST, PGCN_BACKEND, PGCN_SEED.
"""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass


@dataclass
class Tuning:
    # ---- gather kernel (spmm_tasks_kernel) plan ---------------------------------------------------------
    spmm_chunk: int = 1024           # entries per task of a long row ...
    spmm_adaptive_chunk: bool = True # ... shrunk on small blocks (entries / 8192, >= 64): a task is a latency chain
    spmm_small_row: int = 96         # rows up to this many entries are ONE unsliced task
    spmm_affine_small: bool = True   # r05: an unsliced short row's task runs on the XCD of its FULLEST col % 8 slice instead of
                                     # round-robin (those reads meet the rows that XCD's sliced tasks keep in its L2).  Gather part, kernels one
                                     # after the other: Reddit shape 703-706 -> 696 us, SBM 1 975 -> 1 963, products 4 788 -> 4 743
                                     # (profiles/r05_affine_small.txt): ~1 %, the same sign on all three graphs
    group_min_row: int = 4096        # column groups (explicit `ngroups` only) cut rows at least this long
    fpass: str = "auto"              # 64-feature passes of the gather part: auto (whole graphs, f > 64) | 64 | 0
    xcd_swizzle: bool = True         # unsliced plans: one contiguous row range per XCD
    # ---- XCD-sliced storage ---------------------------------------------------------------------------------
    slices: int = 8                  # = XCDs of an MI355X
    slice_min_cols: int = 16384      # narrower operands fit one L2: stored unsliced
    # ---- tiled parts ----------------------------------------------------------------------------------------
    tiles: bool = True               # split dense regions off the gather part at all
    core_tau: float = 0.05           # 128 x 128 LDS core: minimum tile fill
    core_emax: int = 0               # entries per core piece (0 = adaptive)
    core_min_nnz: int = 2000000      # a smaller tiled part does not pay for its three extra launches (r03: the 1.7 M-entry
                                     # local block of an 8-way shard runs 0.067 ms gather-only, 0.099 ms tiled)
    core_min_frac: float = 0.1
    dense_bf16x3: bool = True        # r04: 512 x 128 blocks on the bf16 matrix cores at fp32 accuracy (three-plane split, six
                                     # products) instead of the fp32-MFMA tiles
    dense3_tau: float = 0.20         # blocks at least this full (of 65 536) take that path (r04 sweep on the benchmark graph,
                                     # SpMM launch group: off 1.742 ms, 0.12 1.724, 0.16 1.728, 0.20 1.685, 0.26 1.796)
    dense3_tau_banded: float = 0.12  # ... under a grid aligned to the communities of the vertex order (tuning.order_band_min): the blocks of a community
                                     # are evenly ~23 % full on the planted-partition stand-in and the next cluster sits at ~13 % (r06, SBM line: global
                                     # grid 18.0-18.2 ms / epoch, bands at 0.20 17.5-17.6, bands at 0.12 16.9-17.0: profiles/r06_bands_sbm.txt)
    dense3_piece: int = 0            # blocks per piece (0 = adaptive: one round of 256 pieces, between 1 and 8 blocks)
    dense3_min_blocks: int = 400     # matrices with fewer such blocks leave their entries to the strips / the LDS core (a shard of an
                                     # 8-way run: two more launches and 512-row partial blocks for a few dozen blocks: rank 0 of 8,
                                     # epoch 3.33 ms with them, 3.20 without)
    strip: bool = True               # 512 x 128 strip tiles
    strip_min: int = 512             # stored entries that make a strip tile worth staging ...
    strip_layer_min: int = 384       # ... and one more layer (record) of a tile worth it (r02 sweep)
    strip_big_nnz: int = 20000000    # matrices with at least this many entries (whole graphs, halves: the gather part is bound by fabric
    strip_min_big: int = 256         # reads there, 250 B per entry against 44 B in the strips) take sparser tiles and layers: r04
    strip_layer_min_big: int = 192   # re-sweep on the benchmark graph, a flat optimum -- 512/384 1.645 ms, 256/192 1.622 ms; an 8-way
                                     # shard prefers the r02 values (halo group 0.265 vs 0.298 ms); the 28 M-entry blocks of a 2-way run the
                                     # sparse ones (epoch 7.56 vs 7.96 ms)
    strip_min_records: int = 4096    # blocks with fewer records keep the 128 x 128 LDS core instead (r03: shards of an
                                     # 8-way run: 1.1 k records lose, 4.9 k break even, 8.6 k and 14.9 k win 9-15 %)
    strip_pieces: int = 1024         # upper bound of the strip work pieces (records / 64, in multiples of 256 CUs)
    strip_stage_cost: float = 1.0    # staging a panel ~ this many records of work (piece balancing)
    lanes: str = "strip/gather+dense3"  # launch lanes of the producers of one product: the LDS-bound strips on a second stream beside the gather
                                     # part and the bf16 blocks (r04, Reddit shape: one stream 1.610 ms, this 1.545; strips + dense3 beside the
                                     # gather 1.572, three streams 1.57-1.58; lanes confined to disjoint CU ranges 2.9-5.8 ms: profiles/r04_lanes.txt)
    lanes_min_nnz: int = 20000000    # smaller matrices keep one stream
    # ---- vertex order ---------------------------------------------------------------------------------------
    degree_sort: bool = True
    order: str = "auto"              # degree | community | auto (label propagation, kept when it finds structure)
    order_iters: int = 8
    order_min_inside: float = 0.25
    order_max_share: float = 0.125
    order_min_n: int = 4096
    order_band_min: int = 1024       # r06: the 512 x 128 grid of the bf16 blocks restarts at every community start of a community order (bands
                                     # of at least this many vertices; 0 = one global grid)
    # ---- exchange -------------------------------------------------------------------------------------------
    exchange_rounds: int = 2         # boundary lists are cut into this many all-to-all-v rounds
    # ---- dense H.W (stock library GEMMs) ----------------------------------------------------------------------
    gemm_tuning: bool = True         # use the recorded kernel choices for the n x f x f GEMMs of a layer (tunableop/gfx950.csv + cache) ...
    gemm_tunableop: bool = False     # ... through PyTorch's TunableOp, which also TIMES shapes without a record (set-up: + 20-30 s on a
                                     # cold box) instead of replaying the rocBLAS records by solution index (r04 default)
    dense_fused: int = 3             # the dense products of a layer as the package's own bf16-split MFMA kernels (gemm/pgcn_dense.hip,
                                     # pgcn_wgrad.hip; fp32 accuracy) instead of library GEMMs + ReLU / mask passes: 1 relu(x . W^T); 2 also
                                     # (g (.) mask) . W, the mask travelling as 1 bit per element; 3 also the weight gradient gm^T . x (r06);
                                     # 0 = the library route.  r05 on the MI355X at n = 232 965, f = 128: forward 63.3 us against 85 + 35 us
                                     # of rocBLAS + clamp, input gradient 116.8 us against 50 + 86 us; epochs 10.52 / 10.48 / 10.45 ms off,
                                     # 10.37 / 10.36 / 10.34 at level 2 (profiles/r05_dense_fused_epochs.txt); r06: DESIGN.md section 4
    # ---- GAT path -------------------------------------------------------------------------------------------
    gat_long_row: int = 1024         # rows above this get a 256-thread workgroup in the attention kernels
    gat_small_row: int = 192         # gather plan of the attention structures (r05): a partial row there is heads x d = 1 KB wide, twice the GCN
    gat_chunk: int = 8192            # path's, and every task of a split row leaves one: rows are sliced over the XCDs from 193 entries on (GCN: 97)
                                     # and long rows cut into pieces of 8 192 (GCN: 1 024).  Reddit shape, 4 heads x 64, ms per epoch: 96 / 1 024
                                     # 58.7; 192 / 1 024 55.5; 384 / 1 024 56.4; 192 / 2 048 54.2; 192 / 4 096 53.2; 192 / 8 192 52.8;
                                     # 192 / 16 384 52.8; 320 / 8 192 52.9 (tools/probes_r05/p20_gat_plan.sh)
    gat_blocks: bool = True          # r06: the dense 512 x 128 blocks of the attention pattern on the bf16 matrix cores (pgcn_gat_blocks.hip), weights
    gat_block_tau: float = 0.10      # computed in registers; blocks at least this full (a block costs ~45 us of one CU for four heads, a gathered
    gat_block_piece: int = 0         # blocks per piece of the attention blocks (0 = the adaptive rule of the GCN blocks: one round of 256 pieces, 1..8)
    gat_block_min_frac: float = 0.10 # entry ~18 ns: break-even near 4 %); patterns with less than this fraction of their entries in blocks stay gather-only
    gat_sliced: bool = True          # XCD-sliced edge gradient
    gat_task_grad: bool = True       # edge gradient over the SpMM plan's balanced tasks
    gat_multihead: bool = True       # all heads of attention @ Z in one launch
    gat_fused_grad: bool = True      # edge gradient inside the transposed product's gather pass (one pass fewer)
    gat_fused_forward: bool = True   # forward product with recomputed weights + second accumulator (no alpha / de arrays)


def _parse(spec: str, base: Tuning) -> Tuning:
    fields = {f.name: f for f in dataclasses.fields(Tuning)}
    out = dataclasses.replace(base)
    for item in filter(None, (x.strip() for x in spec.split(","))):
        if "=" not in item:
            raise ValueError("PGCN_TUNING: expected field=value, got %r" % item)
        k, v = (x.strip() for x in item.split("=", 1))
        if k not in fields:
            raise ValueError("PGCN_TUNING: unknown field %r (known: %s)" % (k, ", ".join(sorted(fields))))
        cur = getattr(base, k)
        if isinstance(cur, bool):
            val = v.lower() not in ("0", "false", "no", "off", "")
        elif isinstance(cur, int):
            val = int(v)
        elif isinstance(cur, float):
            val = float(v)
        else:
            val = v
        setattr(out, k, val)
    return out


# what else the package reads from the environment: transports, ingest path, seeds, a cache path, a deadline -- no kernel shape
_OTHER_SWITCHES = {"PGCN_TUNING", "PGCN_EXCHANGE", "PGCN_OVERLAP", "PGCN_INGEST", "PGCN_BACKEND", "PGCN_SEED", "PGCN_DATA_DIR",
                   "PGCN_TUNABLEOP_CACHE", "PGCN_SELFTEST_TIMEOUT", "PGCN_BENCH_BACKEND", "PGCN_BENCH_WATCHDOG", "PGCN_EXTRA_FLAGS",
                   "PGCN_STRIP_PROBE", "PGCN_GATB_PROBE"}


def load(env=None) -> Tuning:
    """The defaults overridden by PGCN_TUNING (read from ``env``, default os.environ).  The per-knob variables of rounds
    1-2 (PGCN_STRIP_PIECES, PGCN_EXCHANGE_ROUNDS, ...) are no longer read: setting one is an error, so that an old probe
    script cannot silently measure the defaults under a label that claims otherwise."""
    env = os.environ if env is None else env
    fields = {f.name.upper() for f in dataclasses.fields(Tuning)}
    stale = sorted(k for k in env if k.startswith("PGCN_") and k not in _OTHER_SWITCHES
                   and (k[5:] in fields or k in ("PGCN_SPMM_FPASS64", "PGCN_ORDER_HUBS", "PGCN_SPMM_PERSIST")))
    if stale:
        raise ValueError("%s: no longer read -- use PGCN_TUNING=\"%s=...\" (tuning.py)" % (", ".join(stale), stale[0][5:].lower()))
    return _parse(env.get("PGCN_TUNING", ""), Tuning())


T = load()
