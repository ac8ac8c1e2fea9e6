#!/bin/bash
# r02 probe 3: L2 hit rate of the gather part with / without column bands
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p3; rm -rf $out; mkdir -p $out
export PGCN_GROUP_MIN_ROW=0
for v in s8c1024k s8g5c1024k s8g8c1024k; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "spmm_tasks" --output-format csv -d $out/$v -- python tools/spmm_probe.py --once $v > $out/$v.log 2>&1 || echo FAILED $v
  echo "== $v"; python tools/pmc_summary.py $out/$v spmm_tasks
done
