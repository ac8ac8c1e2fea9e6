#!/bin/bash
# r03 batch 1: sequential feature passes (1-D grid), dense-tile B prefetch, MALL-served gather ceiling, rank-8 shape
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p1; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "spmm" > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
python tools/spmm_probe.py --variants s8c1024k_p64,s8c1024k_p64s,s8c1024k_p32,s8c1024k_p32s,s8c1024k --split --check > $out/probe_fpass.txt 2>&1
grep -E "median|split|diff" $out/probe_fpass.txt
python tools/spmm_probe.py --variants s8c1024k_p64 --libs base,dpf --split --check > $out/probe_dpf.txt 2>&1
grep -E "median|split|diff" $out/probe_dpf.txt
python tools/spmm_probe.py --workload uniform:232965 --variants s8c1024,s8c1024_p64,s8c1024_p64s,s8c1024_p32s --rounds 5 > $out/probe_uniform.txt 2>&1
grep -E "median" $out/probe_uniform.txt
for v in s8c1024k_p64s s8c1024k_p32s; do
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo "$set" | tr ' ' '+')
    rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/pmc_$v/$t -- python tools/spmm_probe.py --once $v > $out/pmc_${v}_$t.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_$v spmm > $out/pmc_summary_$v.txt; grep -v kernel_trace $out/pmc_summary_$v.txt | grep -A3 tasks_kernel | head -12
done
python tools/rank_probe.py --world 8 > $out/rank8.txt 2>&1; tail -5 $out/rank8.txt
