# VERDICT r04 item 2, third lever: XCD-affine placement of the unsliced short rows (tuning.spmm_affine_small): a short row's task runs on the XCD
# of its fullest col % 8 slice instead of round-robin.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p10; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1$3" | tr '/+ =,-' '_____.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']
print('%-22s %-28s'%('[$1]','$3'), 'ms/epoch %.3f'%r['ms_per_step'], 'group fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), {k:int(v) for k,v in ro.get('split_us',{}).items() if isinstance(v,(int,float))})" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2; do for t in "spmm_affine_small=0" "spmm_affine_small=1"; do run "$t" $rep ""; done; done
for t in "spmm_affine_small=0" "spmm_affine_small=1"; do run "$t" 1 "--generator sbm"; run "$t" 1 "--workload products"; done
