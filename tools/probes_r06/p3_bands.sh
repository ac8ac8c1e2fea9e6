# r06: the block grid of the bf16 blocks aligned to the communities of the vertex order (tuning.order_band_min) on the planted-
# partition stand-in, against the global grid; the GPU tests of the bf16 blocks; kernel stats of the default bench line.
# gpurun --timeout 1500 -- 'bash tools/probes_r06/p3_bands.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p3; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -q -x -k "bf16x3 or dense3 or lanes or strip" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() { n=$(echo "$2" | tr '/+ =,' '_-__.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 400 python bench.py $1 --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); rf=r['roofline']; print('%-44s'%'[$1 $2]', 'ms/epoch %.3f'%r['ms_per_step'], 'group %.4f'%rf['avg_launch_ms'], 'frac %.4f'%rf['frac'], 'split', rf.get('split_us'), '|', rf['kernel'][-260:-150])" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2; do
run "--generator sbm" "order_band_min=0" $rep
run "--generator sbm" "order_band_min=1024" $rep
run "--generator sbm" "order_band_min=1024,dense3_tau=0.12" $rep
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/prof_stdout.log 2> $out/prof_stderr.log
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
f=$(ls $out/prof/*/*kernel_stats.csv $out/prof/*kernel_stats.csv 2>/dev/null | head -1); head -25 "$f" | cut -c1-200
