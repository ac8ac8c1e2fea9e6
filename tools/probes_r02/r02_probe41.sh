#!/bin/bash
# per-rank compute of 2- / 4- / 8-way shards with the second-generation strips: where should the strips stop?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p41; rm -rf $out; mkdir -p $out
for cfg in "2 0" "2 1000000" "4 0" "4 1000000" "8 0" "8 1000000"; do set -- $cfg
  echo "== world $1 min_records $2"; PGCN_STRIP_MIN_RECORDS=$2 python tools/rank_probe.py --world $1 --rank 0 2>&1 | grep -E "forward|backward|rounds=" | cut -c1-260 | tee -a $out/rank_probe.txt
done
