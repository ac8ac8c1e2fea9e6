#!/bin/bash
# r03 batch 13: set-up GEMM tuning in the product (bench N = 1, emulated rank, 2 ranks over gloo); MFMA utilisation of the dense-tile kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p13; rm -rf $out; mkdir -p $out
for t in on off on off; do
  tun=""; [ $t = off ] && tun="gemm_tuning=0"
  PGCN_TUNING="$tun" python bench.py --steps 15 --warmup 3 --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python -c "
import json; r=json.load(open('$out/b_$t.json')); print('gemm tuning $t: ms/epoch %.3f  spmm %.4f  setup %.2f s  loss %.6f  %s' % (r['ms_per_step'], r['roofline']['avg_launch_ms'], r['setup_s'], r['loss'], r['config']['dense_gemm'][:40]))" || tail -3 $out/b_$t.err
done
python bench.py --emulate-rank 0/8 --steps 10 --warmup 2 --no-cpu-baseline > $out/b_r8.json 2>/dev/null; python -c "
import json; r=json.load(open('$out/b_r8.json')); print('rank 0/8 ms/epoch %.3f' % r['ms_per_step'])"
PGCN_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload mid --steps 3 --warmup 1 > $out/b_mid2.json 2> $out/b_mid2.err; tail -c 300 $out/b_mid2.json; echo
ls /tmp/pgcn_tunableop* tunableop* 2>/dev/null | head
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --kernel-include-regex "dense" --output-format csv -d $out/pmc_mfma -- python tools/group_probe.py > $out/pmc_mfma.log 2>&1
python tools/pmc_summary.py $out/pmc_mfma dense
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "run_matches or run_multi" 2>&1 | tail -2
