# r06: the entries outside the blocks as TWO gather structures -- the first N (hub) columns, whose 1 KB rows fit the eight L2s, and the rest
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p18; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 600 python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -5 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1]))
print('%-54s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'], 'loss %.9f' % r['loss'], {k: (round(v, 3) if v else v) for k, v in r['roofline']['pass_split_ms'].items()})
PY
}
for t in "" "gat_hub_cols=8192" "gat_hub_cols=16384" "gat_hub_cols=24576" "gat_hub_cols=49152"; do run "--workload reddit-gat" "$t" 1; done
