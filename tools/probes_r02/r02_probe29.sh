#!/bin/bash
# second-generation strip kernel in the product: strip tests, bench A/B of the layer threshold and the 64-feature passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p29; rm -rf $out; mkdir -p $out
timeout 100 ./tools/micro/strip_bench.bin > $out/strip_bench.txt 2>&1; tail -2 $out/strip_bench.txt
timeout 900 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip or spmm or full_size or engine" > $out/tests.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests.txt | tail -8
run() { # tag, env..., -- bench args
  tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/$tag.err > $out/$tag.json
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'])"
}
run base X=1
run lm320 PGCN_STRIP_LAYER_MIN=320
run lm256 PGCN_STRIP_LAYER_MIN=256 PGCN_STRIP_MIN=384
run p64 PGCN_FPASS=64
