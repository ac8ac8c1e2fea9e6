#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p8; rm -rf $out; mkdir -p $out
echo "== products legacy"
PGCN_STRIP=0 timeout 600 python tools/spmm_probe.py --workload products --rounds 4 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee -a $out/products.txt
for cfg in "128 64" "256 192" "512 384"; do
set -- $cfg
echo "== products strip_min $1 layer_min $2"
PGCN_STRIP_MIN=$1 PGCN_STRIP_LAYER_MIN=$2 timeout 600 python tools/spmm_probe.py --workload products --rounds 4 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee -a $out/products.txt
done
