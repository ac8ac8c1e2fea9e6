#!/bin/bash
# r02 probe 2: column-band (time-sliced column groups) plans for the gather part, every row cut
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p2; rm -rf $out; mkdir -p $out
for cfg in "0 96" "0 24" "256 96"; do
  set -- $cfg
  PGCN_GROUP_MIN_ROW=$1 PGCN_SPMM_SMALL_ROW=$2 timeout 600 python tools/spmm_probe.py --rounds 5 --split \
     --variants s8c1024k,s8g3c1024k,s8g5c1024k,s8g8c1024k,s8g5c1024k0.02 > $out/groups_min$1_small$2.txt 2>&1
  echo "== group_min_row $1 small_row $2"; grep -v amdgpu.ids $out/groups_min$1_small$2.txt | tail -12
done
