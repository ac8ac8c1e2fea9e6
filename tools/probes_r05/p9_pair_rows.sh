# VERDICT r04 item 2, first lever: 4 tasks (pairs of adjacent XCD slices) instead of 8 for rows with small_row < gather entries <= pair_row
# (tuning.spmm_pair_row).  Parity of the SpMM kernels under the option, then the three bench lines.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p9; rm -rf $out; mkdir -p $out
PGCN_TUNING="spmm_pair_row=256" timeout 400 python -m pytest tests/test_hip_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "spmm or fullsize or engine_forward" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() { n=$(echo "$1$3" | tr '/+ =,-' '_____.' | tr -s '_')_$2
  PGCN_TUNING="$1" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); ro=r['roofline']
print('%-22s %-28s'%('[$1]','$3'), 'ms/epoch %.3f'%r['ms_per_step'], 'group fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), {k:int(v) for k,v in ro.get('split_us',{}).items() if isinstance(v,(int,float))})" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2; do for t in "spmm_pair_row=0" "spmm_pair_row=256" "spmm_pair_row=160"; do run "$t" $rep ""; done; done
for t in "spmm_pair_row=0" "spmm_pair_row=256"; do run "$t" 1 "--generator sbm"; run "$t" 1 "--workload products"; done
