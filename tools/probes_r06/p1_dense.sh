# r06, first GPU call: the rewritten dense kernels (gemm/pgcn_dense.hip: column-block-outer rolling pipeline + sign mask;
# gemm/pgcn_wgrad.hip: the weight gradient) beside the r05 kernels on the same data, their GPU tests, and the epoch.
# gpurun --timeout 900 -- 'bash tools/probes_r06/p1_dense.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p1; rm -rf $out; mkdir -p $out
timeout 120 tools/micro/dense_fused_bench.bin > $out/harness.txt 2>&1; echo "harness rc=$?" | tee -a $out/harness.txt
python - <<'PY'
import json
for l in open('gpurun_out/r06_p1/harness.txt'):
    if l.startswith('{'):
        r = json.loads(l)
        if r['forward_us'] > 0:
            print('n=%d f=%3d ok=%-5s fwd %6.1f us (no mask %6.1f; r05 %6.1f)  input grad %6.1f us (no Gm %6.1f; r05 %6.1f)  wgrad %6.1f us (f32mfma %6.1f)  err %.2e %.2e %.2e' % (
                r['n'], r['fin'], r['ok'], r['forward_us'], r['forward_no_mask_us'], r['r05_forward_us'], r['input_grad_us'], r['input_grad_no_gm_us'],
                r['r05_input_grad_us'], r['weight_grad_us'], r['weight_grad_f32mfma_us'], r['err_forward'], r['err_input_grad'], r['err_weight_grad']))
        elif not r['ok']:
            print('MISMATCH', l.strip())
        else:
            print('ok   n=%d fin=%d fout=%d err %.2e %.2e %.2e' % (r['n'], r['fin'], r['fout'], r['err_forward'], r['err_input_grad'], r['err_weight_grad']))
PY
timeout 400 python -m pytest tests/test_zz_dense_fused.py -m gpu -q -x > $out/pytest.txt 2>&1; tail -5 $out/pytest.txt
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')_$2
  PGCN_TUNING="$1" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); print('%-20s'%'[$1]', 'ms/epoch %.3f'%r['ms_per_step'], 'loss', r.get('loss'), 'group ms', r['roofline']['avg_launch_ms'], '|', r['config'].get('dense_fused'))" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2; do for t in "dense_fused=3" "dense_fused=2"; do run "$t" $rep; done; done
