# the bf16-block fill threshold once more on the final r05 tree (r04's sweep had 0.16 ahead of the default 0.20 inside the noise)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p18; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1$3" | tr '/+ =,-' '_____.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']
print('%-22s %-20s'%('[$1]','$3'), 'ms/epoch %.3f'%r['ms_per_step'], 'group fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), {k:int(v) for k,v in ro.get('split_us',{}).items() if isinstance(v,(int,float))})" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2 3; do for t in "dense3_tau=0.20" "dense3_tau=0.16" "dense3_tau=0.18"; do run "$t" $rep ""; done; done
for t in "dense3_tau=0.20" "dense3_tau=0.16"; do run "$t" 1 "--generator sbm"; done
