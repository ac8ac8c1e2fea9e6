#!/usr/bin/env python3
"""Launcher with the reference's name and command line:

    python PGCN.py -a A.mtx -p partvec -b nccl|gloo -s ngpu -l layers -f hidden

(see /root/reference/GPU/PGCN.py:256-287, README.md:85-107).  Everything lives in the
package ``scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd``.
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
_impl = importlib.import_module(
    "scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd.PGCN")
if __name__ != "__main__":
    # alias, not a copy: `import PGCN; PGCN.device = ...` must reach the implementation's module globals
    # (the reference's variants poke them, GPU/PGCN.py:23-35)
    sys.modules[__name__] = _impl

if __name__ == "__main__":
    _impl.main(sys.argv[1:])
