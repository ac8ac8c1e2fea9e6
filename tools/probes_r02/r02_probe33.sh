#!/bin/bash
# L2 hit rate / fabric bytes of the gather part with whole rows, 64- and 32-feature passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p33; rm -rf $out; mkdir -p $out
for v in s8c1024k s8c1024k_p64 s8c1024k_p32; do
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    tag=$(echo "$set" | tr ' ' '+' | cut -c1-40)
    rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm_tasks" --output-format csv -d $out/$v/$tag -- python tools/spmm_probe.py --once $v > $out/$v.$tag.log 2>&1 || echo "FAILED $v $set"
  done
  echo "== $v"; python tools/pmc_summary.py $out/$v spmm_tasks | grep -v "^$"
done 2>&1 | tee $out/summary.txt
