# r06: blocks per piece of the attention blocks (a piece writes a 512-row partial block per head set whatever it holds) and the threshold between
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p16; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 600 python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -5 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1])); b = r['config'].get('blocks') or {}
print('%-74s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'], 'loss %.9f' % r['loss'], 'on blocks %.3f' % (b.get('entries_on_blocks') or 0),
      {k: (round(v, 3) if v else v) for k, v in r['roofline']['pass_split_ms'].items()}, (b.get('structures') or {}).get('fwd'))
PY
}
for t in "gat_block_tau=0.10" "gat_block_tau=0.10,gat_block_piece=16" "gat_block_tau=0.10,gat_block_piece=32" "gat_block_tau=0.08" "gat_block_tau=0.08,gat_block_piece=16" "gat_block_tau=0.12,gat_block_piece=16" "gat_block_tau=0.15"; do run "--workload reddit-gat" "$t" 1; done
