#!/bin/bash
# MFMA threshold against the faster strips
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p30; rm -rf $out; mkdir -p $out
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/$tag.err > $out/$tag.json
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'])"
}
run tau25 PGCN_DENSE_TAU=0.25
run tau30 PGCN_DENSE_TAU=0.30
run tau40 PGCN_DENSE_TAU=0.40
run tau15 PGCN_DENSE_TAU=0.15
