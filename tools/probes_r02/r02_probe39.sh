#!/bin/bash
# strip pieces ordered by panel and dealt to the XCDs as contiguous runs (L2 reuse of staged panels) vs longest-first
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p39; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip" > $out/tests.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests.txt | tail -3
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/$tag.err > $out/$tag.json
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline'].get('avg_launch_ms_backward_AT'),r['roofline']['split_us'])"
}
run lpt PGCN_STRIP_ORDER=lpt
run panel PGCN_STRIP_ORDER=panel
run lpt2 PGCN_STRIP_ORDER=lpt
run panel2 PGCN_STRIP_ORDER=panel
