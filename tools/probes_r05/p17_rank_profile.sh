# per-step kernel list of one emulated rank of an 8-rank job (compute side): where a rank's 3 ms go
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p17; rm -rf $out; mkdir -p $out
python bench.py --emulate-rank 0/8 --graph on --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_rank_0_8.json 2> $out/bench.err
python -c "
import json; r=json.load(open('$out/bench_rank_0_8.json')); print('rank 0/8 eager %.3f ms'%r['ms_per_step'], 'replay', (r.get('graph_replay') or {}).get('ms_per_step'), 'A_loc %.3f'%r['roofline']['avg_launch_ms'], [(h['round'], round(h['avg_launch_ms'],3)) for h in r.get('halo_groups',[])])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o rk -- python $GRAFT_REPO_ROOT/bench.py --emulate-rank 0/8 --graph off --steps 9 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$out/prof_stdout.log 2> $GRAFT_REPO_ROOT/$out/prof_stderr.log
cd $GRAFT_REPO_ROOT; rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
