cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p22; rm -rf $out; mkdir -p $out
for rp in 0/2 0/4; do for t in "strip_big_nnz=20000000" "strip_big_nnz=20000000,dense3_min_blocks=100" "strip_big_nnz=5000000,dense3_min_blocks=100"; do tt=$(echo $rp | tr '/' '_')_$t
  PGCN_TUNING="$t" python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rank_$tt.json 2>/dev/null
  python -c "
import json; r=json.load(open('$out/bench_rank_$tt.json')); print('$rp $t', 'ms/epoch %.3f'%r['ms_per_step'], 'loc %.4f'%r['roofline']['avg_launch_ms'], [round(h['avg_launch_ms'],4) for h in r['halo_groups']])"
done; done
