#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p10; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests_full.txt 2>&1; grep -E "passed|failed|error" $out/tests_full.txt | tail -3
PGCN_STRIP=0 python bench.py --no-cpu-baseline --steps 10 2>/dev/null > $out/bench_legacy.json; python -c "
import json;r=json.load(open('$out/bench_legacy.json'));print('legacy',r['ms_per_step'],r['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline --steps 10 2>/dev/null > $out/bench_strip.json; python -c "
import json;r=json.load(open('$out/bench_strip.json'));print('strip',r['ms_per_step'],r['roofline']['avg_launch_ms'])"
python bench.py --no-cpu-baseline --steps 10 --no-kernel-timing 2>/dev/null > $out/bench_strip_nt.json; python -c "
import json;r=json.load(open('$out/bench_strip_nt.json'));print('strip no timing',r['ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_stdout.log 2> $out/prof_stderr.log
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs head -25
