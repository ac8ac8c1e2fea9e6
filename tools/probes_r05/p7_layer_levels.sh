# tuning.layer_fused levels: 0 = PSpMM + separate dense node, 1 = one node, re-associated backward with the lower layer's mask folded
# into the input gradient (finished operands), 2 = + the fix-up as the loader of the products.  Three runs each.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p7; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']
print('%-16s'%'[$1]', 'ms/epoch %.3f'%r['ms_per_step'], 'group %.4f'%ro['avg_launch_ms'], 'loss', r.get('loss'))" || tail -5 "$out/bench_$n.err"; }
for rep in 1 2 3; do for t in "layer_fused=1" "layer_fused=0"; do run "$t" $rep ""; done; done
