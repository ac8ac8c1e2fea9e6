# the recorded rocBLAS kernels replayed by solution index (no TunableOp in the process): tests, then the epoch against the
# default pick and against TunableOp itself
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p31; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_gemm_direct.py tests/test_launch.py -m gpu -q -k "gemm or smoke or single_gpu" 2>&1 | grep -E "passed|failed|rror" | tail -5 | tee $out/pytest.txt
for i in 1 2; do for t in "" "gemm_tuning=0" "gemm_tunableop=1"; do
  PGCN_TUNING="$t" python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_${t}_$i.json" 2> "$out/bench_${t}_$i.err"
  python -c "
import json; r=json.load(open('$out/bench_${t}_$i.json')); ro=r['roofline']; print('[$t]', 'ms/epoch %.3f'%r['ms_per_step'], 'fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro['avg_launch_ms_backward_AT']), 'setup %.1f s'%r['setup_s'], r['setup_stages_s'], r['config']['dense_gemm'][:60], 'loss', r['loss'])" || tail -5 "$out/bench_${t}_$i.err"
done; done
