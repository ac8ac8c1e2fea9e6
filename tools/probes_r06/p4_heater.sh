# r06: do the dense kernels run slower inside an epoch than in the harness because of what runs before them?  The harness with a
# memory-bound "heater" (a 2 GB device-to-device copy, ~0.7 ms) in front of every timed launch, against the plain back-to-back timing.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p4; rm -rf $out; mkdir -p $out
for h in 0 2048 0 2048; do
  PGCN_BENCH_HEATER=$h timeout 200 tools/micro/dense_fused_bench.bin 232965 10 > $out/harness_$h.txt 2>&1
  python - $out/harness_$h.txt $h <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        r = json.loads(l)
        if r['forward_us'] > 0:
            print('heater %5s MB  f=%3d  fwd %6.1f (r05 %6.1f)  input grad %6.1f (no Gm %6.1f; r05 %6.1f)  wgrad %6.1f (f32mfma %6.1f)' % (
                sys.argv[2], r['fin'], r['forward_us'], r['r05_forward_us'], r['input_grad_us'], r['input_grad_no_gm_us'], r['r05_input_grad_us'], r['weight_grad_us'], r['weight_grad_f32mfma_us']))
PY
done
