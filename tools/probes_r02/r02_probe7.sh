#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p7; rm -rf $out; mkdir -p $out
for cfg in "256 256 0" "512 384 0" "1024 512 0" "512 384 1" "1024 512 1" "2048 768 1" "1024 640 1"; do
set -- $cfg
echo "== strip_min $1 layer_min $2 then_core $3"
PGCN_STRIP_MIN=$1 PGCN_STRIP_LAYER_MIN=$2 PGCN_STRIP_CORE=$3 timeout 600 python tools/spmm_probe.py --rounds 4 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee -a $out/sweep.txt
done
