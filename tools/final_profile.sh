#!/bin/bash
# r01 final evidence: GPU tests, default bench, rocprofv3 kernel stats of the bench command, PMC traffic passes.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-r01}
out=gpurun_out/final_$tag; rm -rf $out; mkdir -p $out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $out/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.txt
python bench.py > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_stdout.log 2> $out/prof_stderr.log
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$set" | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/pmc/$t -- python tools/spmm_probe.py --once s8c1024k > $out/pmc_$t.log 2>&1
done
python tools/pmc_summary.py $out/pmc spmm > $out/pmc_summary.txt; cat $out/pmc_summary.txt | grep -v kernel_trace | head -40
