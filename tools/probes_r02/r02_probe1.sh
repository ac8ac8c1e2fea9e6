#!/bin/bash
# r02 probe 1: launch tests + tile-fill threshold sweep of the SpMM (per-kernel split)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p1; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_launch.py -m gpu -x -q 2>&1 | tail -15 > $out/launch_tests.txt
timeout 900 python tools/spmm_probe.py --rounds 6 --split --check \
   --variants s8c1024k,s8c1024k0.03,s8c1024k0.02,s8c1024k0.01,s8c1024k0.005,s8c1024k0.0025 > $out/tau_sweep.txt 2>&1
cp gpurun_out/spmm_probe.json $out/tau_sweep.json
cat $out/launch_tests.txt; cat $out/tau_sweep.txt | tail -30
