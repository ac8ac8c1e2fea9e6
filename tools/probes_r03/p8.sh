#!/bin/bash
# r03 batch 8: column-sweep feasibility with balanced merged rows
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p8; rm -rf $out; mkdir -p $out
python tools/sweep_probe.py --merge 0 --target 256,512,1024,2048,4096,8192 > $out/sweep.txt 2>&1; grep -E "merge|gather" $out/sweep.txt
for t in 512 2048 8192; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "spmm_tasks" --output-format csv -d $out/pmc_t$t -- python tools/sweep_probe.py --once 0 --target $t > $out/pmc_t$t.log 2>&1
  python tools/pmc_summary.py $out/pmc_t$t spmm_tasks > $out/pmc_sum_t$t.txt
  echo "target=$t $(grep -E 'TCC_HIT|TCC_MISS' $out/pmc_sum_t$t.txt | tr -s ' ' | tr '\n' ' ') $(grep 'mean=.*us' $out/pmc_sum_t$t.txt | sed 's/.*| n=/n=/')"
done
