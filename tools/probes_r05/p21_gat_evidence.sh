# the GAT line's evidence after the plan change: PMC traffic record of the dominant pass (same stamp: gat.py is not a kernel source), bench lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p21; rm -rf $out; mkdir -p $out
cp profiles/pmc_traffic.json $out/pmc_traffic.json
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$set" | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm_heads" --output-format csv -d $out/pmc_gat/$t -- python bench.py --workload reddit-gat --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/pmc_gat_$t.log 2>&1
done
python tools/pmc_summary.py $out/pmc_gat spmm_heads > $out/pmc_summary_gat.txt
python tools/make_pmc_traffic.py $out/pmc_summary_gat.txt $out/pmc_traffic.json profiles/r05_pmc_gat.txt reddit-gat rmat 1 256 random gat_grad "spmm_heads_kernel<4, true, true, false>"
rm -rf $out/pmc_gat
cp $out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --workload reddit-gat --steps 5 --warmup 2 > $out/bench_gat.json 2>/dev/null
python bench.py --workload reddit-gat --emulate-rank 0/4 --steps 5 --warmup 2 > $out/bench_gat_rank_0_4.json 2>/dev/null
python -c "
import json; r=json.load(open('$out/bench_gat.json')); print('GAT ms/epoch %.2f'%r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline']['traffic'])"
