# layer_fused = 2 with the hub rows' long slot lists taken out of the folded loader (layer_fused_cap): epochs against level 0
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p8; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_zz_dense_fused.py -m gpu -q -x -k "fused_layer_on_the_engine" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']; fo=ro.get('fixup_folded_into_consumer') or {}
print('%-34s'%'[$1]', 'ms/epoch %.3f'%r['ms_per_step'], 'group %.4f'%ro['avg_launch_ms'], 'loss', r.get('loss'), {k: round(v,4) for k,v in fo.items() if isinstance(v,float)}, {k: round(v,4) for k,v in (fo.get('backward_AT') or {}).items()})" || tail -5 "$out/bench_$n.err"; }
for t in "layer_fused=2" "layer_fused=2,layer_fused_cap=4" "layer_fused=2,layer_fused_cap=6" "layer_fused=2,layer_fused_cap=12" "layer_fused=0" "layer_fused=2"; do run "$t" 1 ""; done
run "layer_fused=2" sbm "--generator sbm"
