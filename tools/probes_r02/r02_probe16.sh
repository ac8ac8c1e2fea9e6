#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p16; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip" 2>&1 | tail -8 | tee $out/tests.txt
timeout 600 python tools/spmm_probe.py --rounds 6 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee $out/strip.txt
for cfg in "256 192" "384 256" "128 96"; do set -- $cfg
PGCN_STRIP_MIN=$1 PGCN_STRIP_LAYER_MIN=$2 timeout 600 python tools/spmm_probe.py --rounds 4 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee -a $out/strip_sweep.txt; done
