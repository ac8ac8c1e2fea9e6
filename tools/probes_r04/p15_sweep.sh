#!/bin/bash
# r04 batch 15: piece counts of the two tall-tile paths now that the bf16 blocks took the dense part of the strips (PGCN_TUNING sweep)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p15; rm -rf $out; mkdir -p $out
for t in "strip_pieces=1024" "strip_pieces=768" "strip_pieces=512" "dense3_piece=8" "dense3_piece=2" "dense3_tau=0.22" "dense3_tau=0.18" "strip_pieces=768,dense3_piece=8"; do
  PGCN_TUNING="$t" python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err
  python -c "
import json; r=json.load(open('$out/bench_$t.json')); print('$t', 'ms/epoch %.3f'%r['ms_per_step'], 'spmm %.4f'%r['roofline']['avg_launch_ms'], 'bwd %.4f'%r['roofline'].get('avg_launch_ms_backward_AT',0), {k:round(v,1) for k,v in r['roofline']['split_us'].items()}, 'setup %.1f'%r['setup_s'])"
done
