#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p15; rm -rf $out; mkdir -p $out
for m in 0 8192 16384 32768 1000000; do
for w in 2 4; do echo "== min_records $m world $w"; PGCN_STRIP_MIN_RECORDS=$m python tools/rank_probe.py --world $w --rank 0 2>&1 | grep -E "forward|backward|epoch" | tee -a $out/rank_probe.txt; done; done
