#!/bin/bash
# r04 batch 11: the bf16 three-plane blocks inside the library -- the new GPU tests, then the bench line at four block-fill
# thresholds and with the path off (PGCN_TUNING), split per kernel.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p11; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "bf16x3 or mfma or strip_tiles or full_size" > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for t in "dense_bf16x3=0" "dense3_tau=0.12" "dense3_tau=0.16" "dense3_tau=0.20" "dense3_tau=0.26"; do
  PGCN_TUNING="$t" python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err
  python -c "
import json; r=json.load(open('$out/bench_$t.json')); print('$t', 'ms/epoch %.3f'%r['ms_per_step'], 'spmm %.4f'%r['roofline']['avg_launch_ms'], 'bwd %.4f'%r['roofline'].get('avg_launch_ms_backward_AT',0), {k:round(v,1) for k,v in r['roofline']['split_us'].items()}, 'setup %.1f'%r['setup_s'])"
done
