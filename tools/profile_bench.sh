#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; summary copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-r01}
mkdir -p gpurun_out/prof_$tag
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$tag/bench_stdout.log 2> gpurun_out/prof_$tag/bench_stderr.log
tail -1 gpurun_out/prof_$tag/bench_stdout.log
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1); echo "stats file: $f"; head -25 "$f"
rm -f gpurun_out/prof_$tag/*/*kernel_trace.csv   # large; the stats summary is what we keep
