# r06: what of the boundary exchange an 8-way rank would SEE -- the emulated rank with every round occupying the
# comm stream for as long as its largest peer segment takes on one xGMI link (bench.py --pace-exchange 153), 2 rounds against 3.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p7; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 400 python bench.py $1 --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -3 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1])); ex = r.get("exchange") or {}
print('%-64s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'])
for tag in ("forward", "backward"):
    for e in ex.get(tag, []):
        print('    %-8s round %d  out %7.2f MB in %7.2f MB  peer %6.2f MB  %.3f ms on the comm stream, exposed %.3f ms  (%.0f GB/s per link)' % (
            tag, e['round'], e['bytes_out'] / 1e6, e['bytes_in'] / 1e6, e['max_peer_bytes'] / 1e6, e['ms'], e['exposed_ms'], e['GBs_per_link']))
if ex.get("allreduce"): print('    allreduce %.3f ms exposed %.3f' % (ex['allreduce']['ms'], ex['allreduce']['exposed_ms']))
PY
}
run "--emulate-rank 0/8" "exchange_rounds=2" 1
run "--emulate-rank 0/8 --pace-exchange 153" "exchange_rounds=2" 1
run "--emulate-rank 0/8 --pace-exchange 153" "exchange_rounds=3" 1
run "--emulate-rank 0/8 --pace-exchange 153" "exchange_rounds=1" 1
run "--emulate-rank 0/4 --pace-exchange 153" "exchange_rounds=2" 1
run "--emulate-rank 0/2 --pace-exchange 153" "exchange_rounds=2" 1
run "--workload reddit-gat --emulate-rank 0/4 --steps 4" "exchange_rounds=2" 1
run "--workload reddit-gat --emulate-rank 0/4 --pace-exchange 153 --steps 4" "exchange_rounds=2" 1
PGCN_OVERLAP=0 run "--workload reddit-gat --emulate-rank 0/4 --pace-exchange 153 --steps 4" "exchange_rounds=2" 2
timeout 900 python -m pytest tests/test_gat_gpu.py tests/test_launch.py -m gpu -q -x -k "multi_rank or rank_of_four or self_launch or gat_two_ranks or emulated" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
