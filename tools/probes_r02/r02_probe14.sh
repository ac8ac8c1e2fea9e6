#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p14; rm -rf $out; mkdir -p $out
for w in 8 4 2; do python tools/rank_probe.py --world $w --rank 0 2>&1 | grep -v amdgpu.ids | tee -a $out/rank_probe.txt; done
PGCN_STRIP=0 python tools/rank_probe.py --world 8 --rank 0 2>&1 | grep -v amdgpu.ids | tee -a $out/rank_probe_legacy.txt
