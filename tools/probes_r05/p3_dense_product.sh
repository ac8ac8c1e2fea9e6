# The harness winner as the library: its GPU tests, then the epoch with the kernels off / forward + input gradient, three runs each.
# gpurun --timeout 600 -- 'bash tools/probes_r05/p3_dense_product.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p3; rm -rf $out; mkdir -p $out
timeout 60 tools/micro/dense_fused_bench.bin > $out/harness.txt 2>&1; echo "harness rc=$?"; grep -o '"n": 232965.*' $out/harness.txt | cut -c1-260
timeout 400 python -m pytest tests/test_zz_dense_fused.py tests/test_gemm_direct.py -m gpu -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); print('%-20s'%'[$1]', 'ms/epoch %.3f'%r['ms_per_step'], 'group %.4f'%r['roofline']['avg_launch_ms'], 'loss', r.get('loss'))" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2 3; do for t in "dense_fused=0" "dense_fused=2" "dense_fused=1"; do run "$t" $rep; done; done
