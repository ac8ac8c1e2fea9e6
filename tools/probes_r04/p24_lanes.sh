# p23 follow-up: which lane arrangement?  (p23: strips alone on a second stream 1.536 ms, dense3 alone 1.548, one stream 1.610.)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p24; rm -rf $out; mkdir -p $out
for t in "strip/gather+dense3" "strip/dense3+gather" "strip/dense3/gather" "strip/gather/dense3" "dense3/strip/gather" "gather/strip/dense3" "gather/dense3/strip" "dense3/gather+strip" "dense3/gather/strip"; do
  n=$(echo $t | tr '/+' '_-')
  PGCN_TUNING="lanes=$t" python bench.py --steps 15 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']; print('[$t]', 'ms/epoch %.3f'%r['ms_per_step'], 'fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro['avg_launch_ms_backward_AT']), 'loss', r['loss'])" || tail -3 "$out/bench_$n.err"
done
