# r06: the chunked row statistics of the GAT path (tests + epoch, on / off), the SBM line with the banded grid at its own threshold,
# the default line again (nothing may move on a degree-ordered graph).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p6; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_gpu.py -m gpu -q -x -k "statistics or forward_product or attention_kernels or full_size" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 600 python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); rf=r.get('roofline') or {}; print('%-60s'%'[$1 $2]', 'ms/epoch %.3f'%r['ms_per_step'], 'loss', r.get('loss'), 'group', rf.get('avg_launch_ms'))" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2; do
run "--workload reddit-gat" "gat_stat_chunk=0" $rep
run "--workload reddit-gat" "gat_stat_chunk=4096" $rep
run "--workload reddit-gat" "gat_stat_chunk=2048" $rep
run "--workload reddit-gat" "gat_stat_chunk=8192" $rep
done
run "--generator sbm" "order_band_min=1024" 1
run "" "order_band_min=1024" 1
