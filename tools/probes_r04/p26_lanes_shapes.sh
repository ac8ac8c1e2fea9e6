# tuning.lanes = "strip/gather+dense3" as the default: bit-identity tests, then the other shapes with and without it
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p26; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -q -k "lanes or capturable or bf16x3" 2>&1 | tail -5 | tee $out/pytest.txt
run() { n=$(echo "$1 $2" | tr '/+ =,' '_-__.' | tr -s '_')
  PGCN_TUNING="$1" python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']; print('[$1 | $2]', 'ms/epoch %.3f'%r['ms_per_step'], 'loc %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), [round(h['avg_launch_ms'],4) for h in (r.get('halo_groups') or [])], 'replay', r.get('graph_replay_ms_per_epoch'), 'loss', r['loss'])" || tail -3 "$out/bench_$n.err"; }
run "" ""
run "strip_pieces=512" ""
run "lanes=strip/dense3+gather" ""
for rp in 0/8 0/4 0/2; do
  run "lanes=" "--emulate-rank $rp --graph"
  run "lanes_min_nnz=0" "--emulate-rank $rp --graph"
done
run "" "--emulate-rank 0/2 --graph"
run "lanes=" "--workload products"
run "" "--workload products"
run "lanes=" "--generator sbm"
run "" "--generator sbm"
