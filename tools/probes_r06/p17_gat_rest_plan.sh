# r06: the gather plan of the attention structures now walks the 53 % of the entries OUTSIDE the blocks (shorter rows: the dense corner is gone) --
# the r05 plan thresholds (gat_small_row 192, gat_chunk 8 192) once more on that structure
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p17; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 600 python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -5 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1]))
print('%-64s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'], {k: (round(v, 3) if v else v) for k, v in r['roofline']['pass_split_ms'].items()})
PY
}
for t in "" "gat_small_row=96" "gat_small_row=384" "gat_small_row=768" "gat_chunk=2048" "gat_chunk=32768" "gat_small_row=384,gat_chunk=32768" "gat_long_row=4096"; do run "--workload reddit-gat" "$t" 1; done
