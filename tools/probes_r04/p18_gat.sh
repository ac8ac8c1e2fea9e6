#!/bin/bash
# r04 batch 18: GAT -- the fused gather passes per group of heads (panel slice fits the Infinity Cache) vs all heads at once
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p18; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_gat_gpu.py -m gpu -x -q -k "head_groups" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for t in "gat_head_group_mb=0" "gat_head_group_mb=160" "gat_head_group_mb=80"; do
  PGCN_TUNING="$t" python bench.py --workload reddit-gat --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err
  python -c "
import json; r=json.load(open('$out/bench_$t.json')); ro=r.get('roofline') or {}; print('$t', 'ms/epoch %.2f'%r['ms_per_step'], 'dominant %.3f ms'%ro.get('avg_launch_ms',0), r.get('ms_per_layer_fwd_bwd'))" || tail -3 $out/bench_$t.err
done
