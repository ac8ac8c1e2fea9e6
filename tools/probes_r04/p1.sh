#!/bin/bash
# r04 batch 1 (prepared at the end of r03, when the round's GPU budget was spent; NOT yet run):
#   a. the ten GPU tests that hold the HIP engine to the outputs of the reference's own main.c
#      (tests/test_reference_grbgcn.py -- written and run on the CPU box with the numpy stand-in kernels only);
#   b. the dense3 harness with its per-wave phase timers (PGCN_DENSE3_PROBE=3 build: where do the 7.75 us per tile go?
#      tools/micro/mfma_rate says the MFMA dependency pattern is free and one LDS operand read per two MFMAs costs
#      43 instead of 32 cycles);
#   c. the default bench line as the round's starting point.
# usage: gpurun --timeout 900 -- 'bash tools/probes_r04/p1.sh'      (about 3 GPU-minutes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p1; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_reference_grbgcn.py tests/test_hip_gpu.py -m gpu -x -q -k "reference or pargcn" > $out/pytest_ref.txt 2>&1; tail -3 $out/pytest_ref.txt
bash tools/experiments/dense3/build.sh > $out/dense3_build.txt 2>&1 && timeout 120 tools/experiments/dense3/dense3_bench.bin > $out/dense3_bench.txt 2>&1; grep -E "phase timers|6 tiles/piece" $out/dense3_bench.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; r=json.load(open('$out/bench_n1.json')); print('N=1 ms/epoch', r['ms_per_step'], 'spmm', r['roofline']['avg_launch_ms'], r['roofline']['split_us'])"
