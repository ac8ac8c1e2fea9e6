"""MI355X-native drop-in for the aggregation hot path of GPU/PGCN.py.

Layout (only what the path needs):
  csrc/        hand-written gfx950 HIP kernels + the C ABI (include/pgcn_hip.h)
  _lib.py      ctypes binding of libpgcn_hip.so (no fallback)
  kernels.py   typed wrappers taking torch CUDA tensors (device memory / streams only)
  partition.py 1D vertex partition -> local CSR pieces + boundary maps
  engine.py    forward/backward aggregation with overlapped RCCL all-to-all-v
  PGCN.py      mirror of the reference's CLI and layer API
  pargcn.py    the same path driven with Parallel-GCN/main.c's training semantics
  synth.py     seeded synthetic graphs of the benchmark shapes

The directory name is the project's name and is not a Python identifier; import it
with ``importlib.import_module("scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd")``
or through the root-level ``PGCN.py`` launcher.
"""
__all__ = ["PGCN", "engine", "kernels", "partition", "synth", "pargcn"]
