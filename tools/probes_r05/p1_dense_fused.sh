# First GPU call of r05 (≈ 3 minutes of box time): the dense matrix-core kernels of gemm/pgcn_dense.hip.
#   1. the harness with every compiled variant (build it first, here: tools/micro/build_dense_fused_bench.sh): which variant is
#      correct and fastest, what the timing probes say;
#   2. the GPU cases of tests/test_zz_dense_fused.py (they have never run);
#   3. the epoch with the kernels off / forward only / forward + input gradient, twice each (run-to-run noise is +- 0.03 ms):
#      switch tuning.dense_fused on only if the epoch says so; equal losses to ~1e-6 relative are expected, not equal bits.
# gpurun --timeout 420 -- 'bash tools/probes_r05/p1_dense_fused.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p1; rm -rf $out; mkdir -p $out
timeout 60 tools/micro/dense_fused_bench.bin > $out/harness.txt 2>&1; echo "harness rc=$?" | tee -a $out/harness.txt
python - <<'PY'
import json
for l in open('gpurun_out/r05_p1/harness.txt'):
    if l.startswith('{'):
        r = json.loads(l)
        if r['forward_us'] > 0:
            print('%-86s f=%3d ok=%-5s fwd %6.1f us  input grad %6.1f us (no Gm %6.1f)' % (r['variant'][:86], r['fin'], r['ok'], r['forward_us'], r['input_grad_us'], r['input_grad_no_gm_us']))
        elif not r['ok']:
            print('MISMATCH', l.strip())
PY
PGCN_TEST_UNRUN=1 timeout 300 python -m pytest tests/test_zz_dense_fused.py -m gpu -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); print('%-20s'%'[$1]', 'ms/epoch %.3f'%r['ms_per_step'], 'loss', r.get('loss'), '|', r['config'].get('dense_fused'))" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2; do for t in "dense_fused=0" "dense_fused=1" "dense_fused=2" "dense_fused=3"; do run "$t" $rep; done; done
