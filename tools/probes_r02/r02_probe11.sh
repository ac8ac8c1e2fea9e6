#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p11; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_launch.py -m gpu -x -q > $out/tests.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests.txt | tail -12
for g in rmat sbm; do
python bench.py --generator $g --no-cpu-baseline --steps 10 2>$out/bench_$g.err > $out/bench_$g.json; python -c "
import json;r=json.load(open('$out/bench_$g.json'));print('$g',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'],r['config']['vertex_order'],r['setup_s'])"
done
PGCN_ORDER=degree python bench.py --generator sbm --no-cpu-baseline --steps 10 2>/dev/null > $out/bench_sbm_degree.json; python -c "
import json;r=json.load(open('$out/bench_sbm_degree.json'));print('sbm degree-order',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'])"
