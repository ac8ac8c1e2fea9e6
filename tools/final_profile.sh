#!/bin/bash
# Round evidence: GPU tests, smoke, default bench, rocprofv3 kernel stats of the bench command, PMC traffic passes,
# second shapes (products, SBM, GAT).  usage: bash tools/final_profile.sh r02   (under gpurun)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-r02}
out=gpurun_out/final_$tag; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $out/pytest_gpu_full.txt | tail -3 | tee $out/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.txt
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_stdout.log 2> $out/prof_stderr.log
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$set" | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/pmc/$t -- python tools/spmm_probe.py --once s8c1024k_p64 > $out/pmc_$t.log 2>&1
done
python tools/pmc_summary.py $out/pmc spmm > $out/pmc_summary.txt; grep -v kernel_trace $out/pmc_summary.txt | head -40
python tools/make_pmc_traffic.py $out/pmc_summary.txt $out/pmc_traffic.json profiles/${tag}_pmc_final.txt
python bench.py --workload products --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_products.json 2>/dev/null
python bench.py --generator sbm --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_sbm.json 2>/dev/null
python bench.py --workload reddit-gat --steps 5 --warmup 2 > $out/bench_gat.json 2>/dev/null
for f in products sbm gat; do python -c "
import json;r=json.load(open('$out/bench_$f.json'));print('$f', round(r['ms_per_step'],3), r['roofline']['avg_launch_ms'], r['roofline'].get('split_us'))"; done
