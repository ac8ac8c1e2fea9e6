# the panel split of the bf16 blocks issued in front of the gather part on its launch lane (tuning.dense3_split_first)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p19; rm -rf $out; mkdir -p $out
timeout 400 python -m pytest tests/test_hip_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "spmm or fullsize or engine_forward or graph" > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
run() { n=$(echo "$1$3" | tr '/+ =,-' '_____.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']
print('%-26s %-20s'%('[$1]','$3'), 'ms/epoch %.3f'%r['ms_per_step'], 'group fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), {k:int(v) for k,v in ro.get('split_us',{}).items() if isinstance(v,(int,float))})" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2 3; do for t in "dense3_split_first=0" "dense3_split_first=1"; do run "$t" $rep ""; done; done
for t in "dense3_split_first=0" "dense3_split_first=1"; do run "$t" 1 "--generator sbm"; done
