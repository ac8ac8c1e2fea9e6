#!/bin/bash
# r04 batch 19: strip thresholds re-swept (the gather part moves 250 B of fabric traffic per entry, the strips 44): how sparse
# may a layer / a tile be before it is cheaper in the gather kernel?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p19; rm -rf $out; mkdir -p $out
for t in "strip_layer_min=384" "strip_layer_min=256" "strip_layer_min=192" "strip_layer_min=128" "strip_layer_min=256,strip_min=256" "strip_layer_min=192,strip_min=256" "strip_layer_min=128,strip_min=128" "strip_layer_min=256,strip_pieces=768"; do
  PGCN_TUNING="$t" python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err
  python -c "
import json; r=json.load(open('$out/bench_$t.json')); print('$t', 'ms/epoch %.3f'%r['ms_per_step'], 'spmm %.4f'%r['roofline']['avg_launch_ms'], 'bwd %.4f'%r['roofline'].get('avg_launch_ms_backward_AT',0), {k:round(v,1) for k,v in r['roofline']['split_us'].items()}, r['roofline']['kernel'][120:200])"
done
