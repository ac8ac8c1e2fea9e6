#!/bin/bash
# r03 batch 12: the stock GEMMs of the epoch (9 x 7.6 GFLOP, 71-114 us each): rocBLAS vs hipBLASLt, TunableOp
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p12; rm -rf $out; mkdir -p $out
run() { tag=$1; shift
  env "$@" python bench.py --steps 15 --warmup 4 --no-cpu-baseline > $out/b_$tag.json 2> $out/b_$tag.err
  python -c "
import json; r=json.load(open('$out/b_$tag.json')); print('%-28s ms/epoch %.3f  spmm %.4f  non-spmm %.3f' % ('$tag', r['ms_per_step'], r['roofline']['avg_launch_ms'], r['ms_per_step'] - 3*r['roofline']['avg_launch_ms'] - 3*r['roofline']['avg_launch_ms_backward_AT']))" || tail -3 $out/b_$tag.err
}
run base X=0
run rocblas TORCH_BLAS_PREFER_HIPBLASLT=0
run hipblaslt TORCH_BLAS_PREFER_HIPBLASLT=1
run tunable PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$out/tunableop.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=60
run base2 X=0
ls $out; cat $out/tunableop*.csv 2>/dev/null | head -20
