# Where the GAT epoch goes outside its two gather passes: full rocprofv3 kernel stats of the GAT bench command
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p12; rm -rf $out; mkdir -p $out
python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat.json 2> $out/bench_gat.err
python -c "
import json; r=json.load(open('$out/bench_gat.json')); print('GAT ms/epoch %.2f'%r['ms_per_step'], 'dominant pass %.3f ms'%r['roofline']['avg_launch_ms'])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o gat -- python $GRAFT_REPO_ROOT/bench.py --workload reddit-gat --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$out/prof_stdout.log 2> $GRAFT_REPO_ROOT/$out/prof_stderr.log
cd $GRAFT_REPO_ROOT; rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
f=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1); head -60 "$f" | cut -c1-200 > $out/gat_kernel_stats_head.csv; wc -l "$f"
