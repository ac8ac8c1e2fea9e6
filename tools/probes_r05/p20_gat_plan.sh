# GAT: the gather plan's row-slicing threshold for 1 KB rows (a partial row costs twice what it costs the GCN path)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p20; rm -rf $out; mkdir -p $out
for t in "spmm_small_row=192,spmm_chunk=2048" "spmm_small_row=256,spmm_chunk=2048" "spmm_small_row=128,spmm_chunk=2048" "spmm_small_row=192,spmm_chunk=4096" "spmm_small_row=256,spmm_chunk=4096"; do n=$(echo "$t" | tr "=," "__")
  PGCN_TUNING="$t" python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_$n.json 2> $out/bench_$n.err
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); print('%-40s'%'[$t]', 'GAT ms/epoch %.2f'%r['ms_per_step'], 'dominant pass %.3f ms'%r['roofline']['avg_launch_ms'])" || tail -3 $out/bench_$n.err
done
