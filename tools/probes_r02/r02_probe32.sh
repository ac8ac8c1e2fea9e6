#!/bin/bash
# gather part: unsliced-row threshold, 64-feature passes on the products shape
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p32; rm -rf $out; mkdir -p $out
run() { tag=$1; shift; args=""; while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs="$envs $1"; shift; done; shift
  env $envs python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>$out/$tag.err > $out/$tag.json; envs=""
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'])"
}
run sr192 PGCN_SPMM_SMALL_ROW=192 --
run sr384 PGCN_SPMM_SMALL_ROW=384 --
run prod_p0 X=1 -- --workload products
run prod_p64 PGCN_FPASS=64 -- --workload products
run sbm_p64 PGCN_FPASS=64 -- --generator sbm
