#!/bin/bash
# r03 batch 14: MFMA-tile pieces (the dense kernel's matrix pipes are 56 % busy: SQ_VALU_MFMA_BUSY_CYCLES) -- longer pieces, lower threshold
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p14; rm -rf $out; mkdir -p $out
run() { tag=$1; tun=$2
  PGCN_TUNING="gemm_tuning=0,$tun" python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/b_$tag.json 2> $out/b_$tag.err
  python -c "
import json; r=json.load(open('$out/b_$tag.json')); print('%-24s ms/epoch %.3f  spmm %.4f bwd %.4f %s' % ('$tag', r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['avg_launch_ms_backward_AT'], {k: round(v) for k, v in r['roofline']['split_us'].items()}))" || tail -3 $out/b_$tag.err
}
run base ""
run piece6 "dense_piece=6"
run piece12 "dense_piece=12"
run piece6_tau25 "dense_piece=6,dense_tau=0.25"
run piece12_tau25 "dense_piece=12,dense_tau=0.25"
run base2 ""
