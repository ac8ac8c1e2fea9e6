# r06: is a whole training step replayed from one HIP graph faster than the eager launches at N = 1 (fork / join of the launch lanes,
# ~70 kernel boundaries)?  And one stream against two lanes under replay.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p10; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 400 python bench.py $1 --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -3 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1])); g = r.get("graph_replay") or {}
print('%-50s' % ('[' + sys.argv[2] + ']'), 'eager ms/epoch %.3f' % r['ms_per_step'], 'group %.4f' % r['roofline']['avg_launch_ms'], 'replay', g.get('ms_per_step'), {k: v for k, v in g.items() if k in ('captured', 'error', 'nodes')})
PY
}
for rep in 1 2; do
run "--graph on" "" $rep
run "--graph on" "lanes=" $rep
done
