#!/bin/bash
# the GAT file after the wrapper fix (all tests, no -x) + kernel stats of the GAT bench line (what is left outside the
# three gather passes?)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p20; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_gpu.py -q -m gpu --durations=8 > $out/pytest_gat.txt 2>&1; tail -16 $out/pytest_gat.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o gat -- python bench.py --workload reddit-gat --steps 4 --warmup 1 --no-cpu-baseline > $out/prof_stdout.log 2> $out/prof_stderr.log
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
f=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:45]:
    print("%-100s %6s avg %9.1f us  total %9.1f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
