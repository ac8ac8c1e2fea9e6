# r06: (a) does the row stride of the feature panel cost the gather part L2 capacity (tools/layout_probe.py)?
#      (b) the mid line (n = 131 072, f = 64, 2 layers) at the dense levels 0 / 2 / 3: 0.71 ms in r05, 0.774 in the r06 evidence run.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p11; rm -rf $out; mkdir -p $out
timeout 600 python tools/layout_probe.py --reps 10 2> $out/layout.err | tee $out/layout.txt; tail -3 $out/layout.err
timeout 600 python tools/layout_probe.py --generator sbm --reps 10 2> $out/layout_sbm.err | tee $out/layout_sbm.txt; tail -3 $out/layout_sbm.err
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 400 python bench.py $1 --steps 40 --warmup 5 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -3 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1]))
print('%-50s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.4f' % r['ms_per_step'], 'group %.4f' % r['roofline']['avg_launch_ms'])
PY
}
for rep in 1 2 3; do
for lv in 0 2 3; do run "--workload mid" "dense_fused=$lv" $rep; done
done
