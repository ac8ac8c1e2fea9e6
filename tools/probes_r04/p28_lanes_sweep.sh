# with the producers on two streams the balance between the parts may sit elsewhere: lane permutations not yet tried, then the
# thresholds of every part again (one knob at a time around the defaults)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p28; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')
  PGCN_TUNING="$1" python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']; print('%-44s'%'[$1]', 'ms/epoch %.3f'%r['ms_per_step'], 'fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), {k:round(v) for k,v in ro.get('split_us',{}).items()})" || tail -3 "$out/bench_$n.err"; }
for t in "" "lanes=dense3/strip+gather" "lanes=strip+gather/dense3" "lanes=gather+strip/dense3" "lanes=dense3+gather/strip" \
  "strip_layer_min_big=128" "strip_layer_min_big=256" "strip_layer_min_big=320" "strip_min_big=192,strip_layer_min_big=160" "strip_min_big=384,strip_layer_min_big=256" \
  "dense3_tau=0.16" "dense3_tau=0.24" "dense3_tau=0.28" "fpass=0" "strip_pieces=768" "strip_pieces=1280" "dense3_piece=4" "dense3_piece=6" \
  "strip_stage_cost=2.0" "strip_stage_cost=0.5" "spmm_adaptive_chunk=0" "xcd_swizzle=0" ""; do run "$t"; done
