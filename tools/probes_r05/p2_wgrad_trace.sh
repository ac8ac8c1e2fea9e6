# Which of the two kernels of gemm/pgcn_wgrad.hip takes its 565 us (p1: 7 x the 64-slab library product)?
# gpurun --timeout 240 -- 'bash tools/probes_r05/p2_wgrad_trace.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p2; rm -rf $out; mkdir -p $out
cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -o wg -- $GRAFT_REPO_ROOT/tools/micro/dense_fused_bench.bin > $GRAFT_REPO_ROOT/$out/harness.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-90s calls %5s avg %10.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
