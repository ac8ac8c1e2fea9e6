#!/bin/bash
# r03 batch 18: vectorised loss kernels; the whole GPU suite on the final sources
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p18; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; grep -E "passed|failed" $out/pytest_gpu.txt | tail -2; grep -E "^FAILED|^ERROR" $out/pytest_gpu.txt | head
for i in 1 2; do python bench.py --steps 15 --warmup 3 --no-cpu-baseline > $out/b_$i.json 2>/dev/null; python -c "
import json; r=json.load(open('$out/b_$i.json')); print('N=1 ms/epoch %.3f spmm %.4f %s' % (r['ms_per_step'], r['roofline']['avg_launch_ms'], {k: round(v) for k, v in r['roofline']['split_us'].items()}))"; done
