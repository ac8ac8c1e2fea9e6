# config 4 (f = 64, partial rows half as wide as the benchmark's): the gather plan's slicing threshold on rank 0 of 8 at full size
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p22; rm -rf $out; mkdir -p $out
python tools/make_shards.py --workload papers --ranks 8 --only-rank 0 --device cuda --out /tmp/papers > $out/make_shards.txt 2>&1
for t in "spmm_small_row=96" "spmm_small_row=48" "spmm_small_row=192"; do n=$(echo "$t" | tr '=,' '__')
  PGCN_TUNING="$t" timeout 300 python bench.py --workload papers --emulate-rank 0/8 --shards /tmp/papers --partvec block --features 64 --layers 2 --steps 4 --warmup 1 --no-cpu-baseline > $out/bench_$n.json 2> $out/bench_$n.err
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); print('%-24s'%'[$t]', 'ms/epoch %.2f'%r['ms_per_step'], 'A_loc %.3f'%r['roofline']['avg_launch_ms'], [round(h['avg_launch_ms'],3) for h in r['halo_groups']])" || tail -3 $out/bench_$n.err
done
