# r06: the attention blocks on a grid that restarts at the bands of the vertex order (as the GCN blocks' does): the planted-partition stand-in
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p19; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_blocks.py -x -q 2>&1 | tail -6 | tee $out/pytest.txt
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 600 python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -5 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1])); b = r['config'].get('blocks') or {}
print('%-62s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'], 'loss %.9f' % r['loss'], 'on blocks %.3f' % (b.get('entries_on_blocks') or 0),
      {k: (round(v, 3) if v else v) for k, v in r['roofline']['pass_split_ms'].items()}, (b.get('structures') or {}).get('fwd'), r['config']['vertex_order'].get('bands'))
PY
}
run "--workload reddit-gat --generator sbm" "gat_blocks=0" 1
run "--workload reddit-gat --generator sbm" "" 1
run "--workload reddit-gat --generator sbm" "order_band_min=0" 1
run "--workload reddit-gat --generator sbm" "gat_block_tau=0.06" 1
run "--workload reddit-gat" "" 1
