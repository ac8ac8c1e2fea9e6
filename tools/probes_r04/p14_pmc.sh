#!/bin/bash
# r04 batch 14: per-kernel L2<->fabric traffic of the A_loc.H launch group with the bf16 blocks in (three --pmc passes of
# tools/group_probe.py, kernel trace only), summary into gpurun_out/r04_p14/.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p14; rm -rf $out; mkdir -p $out
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$set" | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/pmc/$t -- python tools/group_probe.py "$@" > $out/pmc_$t.log 2>&1
done
python tools/pmc_summary.py $out/pmc spmm > $out/pmc_summary.txt
rm -rf $out/pmc
cat $out/pmc_summary.txt
