#!/bin/bash
# automatic feature passes: whole graph on, 8-way shard off
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p43; rm -rf $out; mkdir -p $out
python tools/rank_probe.py --world 8 --rank 0 2>&1 | grep -E "forward|backward" | cut -c1-100 | tee -a $out/rank_probe.txt
python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/base.err > $out/base.json
python -c "
import json;r=json.load(open('$out/base.json'));print('base',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline']['split_us'],r['roofline']['kernel'][:90])"
