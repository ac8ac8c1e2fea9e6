# The whole GPU suite + smoke + the default bench line (what the driver runs at round end).
# gpurun --timeout 2400 -- 'bash tools/probes_r06/p2_full_suite.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p2; rm -rf $out; mkdir -p $out
timeout 2000 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 600 python bench.py > $out/bench_stdout.json 2> $out/bench_stderr.txt; python -c "
import json; r=json.load(open('$out/bench_stdout.json')); print('ms/epoch %.3f'%r['ms_per_step'], 'group', r['roofline']['avg_launch_ms'], 'frac', r['roofline']['frac'], 'cpu', r['cpu_baseline'].get('value'))"
