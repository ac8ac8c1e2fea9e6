#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p12; rm -rf $out; mkdir -p $out
for h in 0 0.02 0.04 0.08 0.15; do
PGCN_ORDER_HUBS=$h python bench.py --generator sbm --no-cpu-baseline --steps 6 2>/dev/null > $out/bench_sbm_h$h.json; python -c "
import json;r=json.load(open('$out/bench_sbm_h$h.json'));print('sbm hubs $h',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'])"
done
