# GAT: the gather plan's row-slicing threshold for 1 KB rows (a partial row costs twice what it costs the GCN path)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p20; rm -rf $out; mkdir -p $out
for t in ""; do n=default
  PGCN_TUNING="$t" python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$n.json 2> $out/bench_$n.err
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); print('%-40s'%'[$t]', 'GAT ms/epoch %.2f'%r['ms_per_step'], 'dominant pass %.3f ms'%r['roofline']['avg_launch_ms'])" || tail -3 $out/bench_$n.err
done
