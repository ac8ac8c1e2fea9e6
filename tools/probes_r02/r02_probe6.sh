#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p6; rm -rf $out; mkdir -p $out
for dbg in 0 1 2 3; do
echo "== dbg $dbg (1 = no compute, 2 = no panel staging)"
PGCN_STRIP_DBG=$dbg timeout 600 python tools/spmm_probe.py --rounds 4 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | grep split | tee -a $out/dbg.txt
done
