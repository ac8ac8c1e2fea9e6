#!/bin/bash
# feature passes of the gather part: 64 / 32 features per pass (grid y, pass-major) vs whole rows
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p19; rm -rf $out; mkdir -p $out
run() { # tag, env, bench args
  tag=$1; shift; fp=$1; shift
  PGCN_FPASS=$fp python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>$out/$tag.err > $out/$tag.json
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'])"
}
run rmat_p0 0
run rmat_p64 64
run rmat_p32 32
run sbm_p0 0 --generator sbm
run sbm_p32 32 --generator sbm
run sbm_p64 64 --generator sbm
run products_p0 0 --workload products
run products_p32 32 --workload products
