#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p9; rm -rf $out; mkdir -p $out
for cfg in "1024 0.20" "512 0.20" "768 0.20" "1024 0.15" "1024 0.12"; do
set -- $cfg
echo "== pieces $1 dense_tau $2"
PGCN_STRIP_PIECES=$1 PGCN_DENSE_TAU=$2 timeout 600 python tools/spmm_probe.py --rounds 4 --split --variants s8c1024k 2>&1 | grep -v amdgpu.ids | tee -a $out/sweep.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $out/tests.txt
python bench.py --no-cpu-baseline 2>/dev/null | tee $out/bench.json
