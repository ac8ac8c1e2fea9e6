"""ctypes binding of libpgcn_hip.so (the C ABI declared in include/pgcn_hip.h).

There is deliberately NO fallback: if the shared library is missing or does not
export a symbol the header declares, importing / calling raises.  The library is
built in-tree by ``csrc/build.sh`` (``__graft_entry__.build()``), never JIT-cached
outside the repository.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

# torch must map its own libamdhip64.so.7 / librccl.so.1 first: libpgcn_hip.so links
# to the same SONAMEs and then shares torch's HIP runtime (streams, allocations).
import torch  # noqa: F401  (import for its side effect on the loader)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpgcn_hip.so")

PGCN_OK = 0
PGCN_EUNSUPPORTED = -5
SPMM_ACCUMULATE = 1
SPMM_XCD_SWIZZLE = 2
SPMM_OFFSETS32 = 4
SPMM_NO_FIXUP = 8
SPMM_FPASS64 = 16
MAX_SLICES = 8

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_u32 = ctypes.c_uint32

# name -> (restype, argtypes); must list EVERY symbol of include/pgcn_hip.h
SIGNATURES = {
    "pgcn_abi_version": (ctypes.c_int, []),
    "pgcn_last_error": (ctypes.c_char_p, []),
    "pgcn_device_info": (ctypes.c_int, [_i32, ctypes.POINTER(_i64)]),
    "pgcn_mtx_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "pgcn_mtx_read_coo": (ctypes.c_int, [ctypes.c_char_p, _i64, _vp, _vp, _vp, ctypes.POINTER(_i64), _i32]),
    "pgcn_build_comm_maps": (ctypes.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _i32]),
    "pgcn_load_mtx_partition": (ctypes.c_int, [ctypes.c_char_p, _vp, _i64, _i32, _i64, _vp, _vp, _vp,
                                               ctypes.POINTER(_i64), _i32]),
    "pgcn_shard_write": (ctypes.c_int, [ctypes.c_char_p, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "pgcn_shard_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_i64)]),
    "pgcn_shard_read": (ctypes.c_int, [ctypes.c_char_p, _i64, _i64, _vp, _vp, _vp, _vp]),
    "pgcn_spmm_csr_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _u32, _vp]),
    "pgcn_spmm_csr_plan_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _vp,
                                              _i64, _vp, _i64, _i32, _vp, _i64, _i64, _u32, _vp]),
    "pgcn_spmm_core_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp,
                                          _i64, _i64, _vp]),
    "pgcn_spmm_strip_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp]),
    "pgcn_spmm_heads_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _i64,
                                           _vp, _i64, _vp, _i64, _i64, _u32, _vp]),
    "pgcn_spmm_heads_recompute_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_float, _i32, _i32, _i32, _i64, _vp, _i64,
                                                     _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _u32, _vp]),
    "pgcn_spmm_heads_grad_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_float, _i32, _i32, _i32, _i64, _vp, _i64,
                                                _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64,
                                                _u32, _vp]),
    "pgcn_spmm_heads_forward2_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_float, _i32, _i32, _i32, _i64, _vp, _i64,
                                                    _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _u32,
                                                    _vp]),
    "pgcn_gat_blocks_forward_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, ctypes.c_float, _i32, _i32, _i64, _i64,
                                                   _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    "pgcn_gat_blocks_backward_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, ctypes.c_float, _i32,
                                                    _i32, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    "pgcn_spmm_dense_bf16x3_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _i64, _vp]),
    "pgcn_dense_bf16x3_image_bytes": (_i64, [_i64, _i32]),
    "pgcn_spmm_fixup_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _u32, _vp]),
    "pgcn_spmm_plan_host": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp,
                                           ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                           ctypes.POINTER(_i64)]),
    "pgcn_spmm_plan_host_ex": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp,
                                              ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                              ctypes.POINTER(_i64)]),
    "pgcn_gat_edge_softmax_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32,
                                                 ctypes.c_float, _i32, _i64, _vp, _vp, _vp, _vp]),
    "pgcn_gat_edge_weights_t_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32,
                                                   ctypes.c_float, _i32, _vp, _vp]),
    "pgcn_gat_edge_grad_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                              _vp, _i64, _vp, _i64, _vp, _i32, _i32, ctypes.c_float, _i32, _vp, _vp, _vp]),
    "pgcn_gat_edge_grad_sliced_f32": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                                     _vp, _i64, _vp, _i64, _vp, _i32, _i32, ctypes.c_float, _i32, _vp, _vp,
                                                     _vp]),
    "pgcn_gat_edge_grad_tasks_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64,
                                                    _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, ctypes.c_float, _i32,
                                                    _vp, _vp, _vp, _i64, _i64, _vp]),
    "pgcn_csr_row_sums_f32": (ctypes.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp]),
    "pgcn_gat_row_dots_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "pgcn_csr_permute_f32": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "pgcn_nll_rows_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "pgcn_nll_rows_backward_f32": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, ctypes.c_float, _i64, _i32, _vp, _i64, _vp]),
    "pgcn_gather_rows_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp]),
    "pgcn_scatter_rows_f32": (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "pgcn_comm_unique_id": (ctypes.c_int, [_vp]),
    "pgcn_comm_init": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, _i32, _i32]),
    "pgcn_comm_destroy": (ctypes.c_int, [_vp]),
    "pgcn_exchange_alltoallv_f32": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "pgcn_allreduce_sum_f32": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
}

_LIB = None


class PgcnError(RuntimeError):
    pass


ABI_VERSION = 3      # PGCN_ABI_VERSION of include/pgcn_hip.h


def build(verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["bash", os.path.join(_HERE, "csrc", "build.sh")], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout + out.stderr)
    if out.returncode != 0:
        raise PgcnError("building libpgcn_hip.so failed")
    # the GEMM plumbing beside it (gemm/pgcn_gemm.cpp: rocBLAS calls by solution index; PGCN.mm_nt / mm_nn fall back to
    # PyTorch's product without it)
    out = subprocess.run(["bash", os.path.join(_HERE, "gemm", "build.sh")], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout + out.stderr)
    if out.returncode != 0:
        raise PgcnError("building libpgcn_gemm.so failed")
    return LIB_PATH


def lib():
    """The loaded library with typed entry points.  Raises if it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PgcnError(
                "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the product path has no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if L.pgcn_abi_version() != ABI_VERSION:
            raise PgcnError("libpgcn_hip.so has ABI version %d, this package binds version %d: rebuild (csrc/build.sh)"
                            % (L.pgcn_abi_version(), ABI_VERSION))
        _LIB = L
    return _LIB


def load_variant(path: str):
    """Load ANOTHER build of the library (A/B probes: tools/ab_build.sh); typed like lib()."""
    L = ctypes.CDLL(path)
    L.pgcn_abi_version.restype = ctypes.c_int
    if L.pgcn_abi_version() != ABI_VERSION:       # same names, other layouts: refuse instead of computing nonsense
        raise PgcnError("%s has ABI version %d, this package binds version %d" % (path, L.pgcn_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name, None)       # a variant build may lack entry points (probe builds)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    return L


def check(rc: int, what: str = "") -> None:
    if rc != PGCN_OK:
        msg = lib().pgcn_last_error().decode("utf-8", "replace")
        raise PgcnError("%s failed (rc=%d): %s" % (what or "libpgcn_hip call", rc, msg))
