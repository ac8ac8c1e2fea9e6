# r06: the GAT engine with its exchanges on the comm stream -- forward split (s2 first, halo columns last),
# backward split (halo rows first) -- multi-rank GPU tests and the paced emulated rank 0 of 4.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p8; rm -rf $out; mkdir -p $out
timeout 1200 python -m pytest tests/test_gat_gpu.py tests/test_launch.py -m gpu -q -x -k "multi_rank or rank_of_four or gat_two_ranks or reference_mode or pgat_cli" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  timeout 400 python bench.py $1 --steps 4 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -3 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1])); ex = r.get("exchange") or {}
print('%-64s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'], 'loss', r['loss'])
for tag in ("forward_s2", "forward", "backward"):
    for e in ex.get(tag, []):
        print('    %-10s round %d  out %7.2f MB  peer %6.2f MB  %.3f ms on the comm stream, exposed %.3f ms' % (tag, e['round'], e['bytes_out'] / 1e6, e['max_peer_bytes'] / 1e6, e['ms'], e['exposed_ms']))
PY
}
run "--workload reddit-gat --emulate-rank 0/4" "" 1
run "--workload reddit-gat --emulate-rank 0/4 --pace-exchange 153" "" 1
PGCN_OVERLAP=0 run "--workload reddit-gat --emulate-rank 0/4 --pace-exchange 153" "overlap0" 1
run "--workload reddit-gat" "" 1
