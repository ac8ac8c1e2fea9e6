# what the TunableOp choices buy per epoch today (p29: 17 + 13 us per layer on random operands, 33 s of set-up on a cold box)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p30; rm -rf $out; mkdir -p $out
for i in 1 2 3; do for t in "gemm_tuning=0" "gemm_tuning=1"; do
  PGCN_TUNING="$t" python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_${t}_$i.json" 2> "$out/bench_${t}_$i.err"
  python -c "
import json; r=json.load(open('$out/bench_${t}_$i.json')); ro=r['roofline']; print('[$t]', 'ms/epoch %.3f'%r['ms_per_step'], 'fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro['avg_launch_ms_backward_AT']), 'setup %.1f s'%r['setup_s'])"
done; done
