# r06: the dense blocks of the attention pattern on the matrix cores (csrc/pgcn_gat_blocks.hip): parity first, then the GAT line with
# and without them and at three block thresholds, then the kernel list of the default.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p12; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_blocks.py -x -q 2>&1 | tail -15 | tee $out/pytest.txt
run() { n=$(echo "$1 $2" | tr '/+ =,-' '_____.' | tr -s '_')_$3
  PGCN_TUNING="$2" timeout 600 python bench.py $1 --steps 5 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" "$1 $2" <<'PY' || tail -5 "$out/bench_$n.err"
import json, sys
r = json.load(open(sys.argv[1]))
print('%-60s' % ('[' + sys.argv[2] + ']'), 'ms/epoch %.3f' % r['ms_per_step'], 'loss', r.get('loss'), r.get('config', {}).get('blocks'))
PY
}
for t in "gat_blocks=0" "" "gat_block_tau=0.04" "gat_block_tau=0.10" "gat_block_tau=0.03"; do run "--workload reddit-gat" "$t" 1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o gat -- python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/prof_stdout.log 2> $out/prof_stderr.log
f=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1); grep -E "Name|spmm_heads|gat_|fixup|split_panels|row_dots" "$f" | cut -c1-200 | head -20
