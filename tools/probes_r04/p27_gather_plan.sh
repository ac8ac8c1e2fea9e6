# the gather part shrank since r02 tuned its plan (30 % of its entries sit in unsliced rows <= 96 entries): re-sweep the plan knobs under lanes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p27; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -q -k "lanes or capturable or bf16x3" 2>&1 | grep -E "passed|failed|error" | tee $out/pytest.txt
run() { n=$(echo "$1 $2" | tr '/+ =,' '_-__.' | tr -s '_')
  PGCN_TUNING="$1" python bench.py $2 --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']; print('[$1 | $2]', 'ms/epoch %.3f'%r['ms_per_step'], 'loc %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), ro.get('split_us'))" || tail -3 "$out/bench_$n.err"; }
for t in "spmm_small_row=32" "spmm_small_row=48" "spmm_small_row=64" "spmm_small_row=128" "spmm_small_row=192" "spmm_chunk=512" "spmm_chunk=2048" "spmm_small_row=64,strip_pieces=512"; do run "$t" ""; done
