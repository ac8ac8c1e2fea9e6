# r06: the block kernel with the weights of step s + 1 built under the MFMAs of step s (32-row waves, two plane sets), and the blocks of the
# split structures (rank 0 of 4 with the exchange priced at the link rate).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p13; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_blocks.py -x -q 2>&1 | tail -8 | tee $out/pytest.txt
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 600 python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -5 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1])); b = r['config'].get('blocks') or {}
print('%-74s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'], 'loss %.9f' % r['loss'], 'on blocks %.3f' % (b.get('entries_on_blocks') or 0),
      {k: (round(v, 3) if v else v) for k, v in r['roofline']['pass_split_ms'].items()})
PY
}
for t in "gat_block_tau=0.10" "gat_block_tau=0.06" "gat_block_tau=0.04" "gat_blocks=0"; do run "--workload reddit-gat" "$t" 1; done
for t in "gat_blocks=0" "gat_block_tau=0.10" "gat_block_tau=0.06"; do run "--workload reddit-gat --emulate-rank 0/4 --pace-exchange 153" "$t" 1; done
run "--workload reddit-gat --emulate-rank 0/4" "gat_block_tau=0.06" 1
