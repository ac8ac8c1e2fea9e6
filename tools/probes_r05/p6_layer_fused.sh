# The fix-up folded into the dense product (tuning.layer_fused): GPU tests of the loader and of the layer node, then the epoch with the
# fused layer on / off, three runs each.
# gpurun --timeout 900 -- 'bash tools/probes_r05/p6_layer_fused.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p6; rm -rf $out; mkdir -p $out
timeout 500 python -m pytest tests/test_zz_dense_fused.py -m gpu -q -x > $out/pytest.txt 2>&1; tail -15 $out/pytest.txt
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']; fo=ro.get('fixup_folded_into_consumer') or {}
print('%-16s'%'[$1]', 'ms/epoch %.3f'%r['ms_per_step'], 'group %.4f'%ro['avg_launch_ms'], 'loss', r.get('loss'), {k: round(v,4) for k,v in fo.items() if isinstance(v,float)}, {k: round(v,4) for k,v in (fo.get('backward_AT') or {}).items()})" || tail -5 "$out/bench_$n.err"; }
for rep in 1 2 3; do for t in "layer_fused=1" "layer_fused=0"; do run "$t" $rep ""; done; done
run "layer_fused=1" sbm "--generator sbm"; run "layer_fused=0" sbm "--generator sbm"
