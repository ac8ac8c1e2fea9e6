#!/bin/bash
# piece count of the strip kernel with the second-generation kernel; new variant tests
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p40; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip or variants" > $out/tests.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests.txt | tail -5
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/$tag.err > $out/$tag.json
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline'].get('avg_launch_ms_backward_AT'),r['roofline']['split_us'])"
}
run p1024 X=1
run p768 PGCN_STRIP_PIECES=768
run p1536 PGCN_STRIP_PIECES=1536
run p512 PGCN_STRIP_PIECES=512
