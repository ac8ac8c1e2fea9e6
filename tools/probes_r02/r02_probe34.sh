#!/bin/bash
# does the gather part need its 24 waves per CU?  (unused dynamic LDS caps the resident workgroups)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p34; rm -rf $out; mkdir -p $out
for pad in 0 38000 51000 78000; do
  PGCN_GATHER_LDS_PAD=$pad python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/pad$pad.err > $out/pad$pad.json
  python -c "
import json;r=json.load(open('$out/pad$pad.json'));print('pad $pad',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'])"
done
