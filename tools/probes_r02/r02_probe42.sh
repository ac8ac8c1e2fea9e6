#!/bin/bash
# P = 8 shard: what slowed it (0.45 ms in r02a)?  MFMA threshold 0.30 vs 0.20 without strips, feature passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p42; rm -rf $out; mkdir -p $out
for cfg in "X=1" "PGCN_DENSE_TAU=0.2" "PGCN_FPASS=0" "PGCN_DENSE_TAU=0.2 PGCN_FPASS=0"; do
  echo "== $cfg"; env $cfg python tools/rank_probe.py --world 8 --rank 0 2>&1 | grep -E "forward|backward" | cut -c1-100 | tee -a $out/rank_probe.txt
done
