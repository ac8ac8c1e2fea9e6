#!/bin/bash
# r02 probe 5: strip kernel -- parity tests, then timing vs the legacy core
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p5; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip or mfma or dense_core or halo_dense" 2>&1 | tail -15 | tee $out/tests.txt
timeout 600 python tools/spmm_probe.py --rounds 6 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee $out/strip.txt
PGCN_STRIP=0 timeout 600 python tools/spmm_probe.py --rounds 6 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee $out/legacy.txt
for m in 64 256 512; do
PGCN_STRIP_MIN=$m timeout 600 python tools/spmm_probe.py --rounds 6 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee $out/strip_min$m.txt
done
