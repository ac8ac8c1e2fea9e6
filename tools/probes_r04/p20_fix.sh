#!/bin/bash
# r04 batch 20: what final_profile.sh r04 found broken (bench's cpu_baseline leg on a device partition, smoke's assertion, one
# test's slot count) re-run; the default bench line with its CPU legs; the 8-way shard with and without bf16 blocks
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p20; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_gpu.py tests/test_launch.py -m gpu -x -q -k "dense_core_lds or single_gpu_line" > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.txt
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 2500 $out/bench.json; echo
python bench.py --workload mid --steps 10 --warmup 2 > $out/bench_mid.json 2>/dev/null; tail -c 600 $out/bench_mid.json; echo
for t in "dense3_min_blocks=0" "dense3_min_blocks=100000" "dense3_min_blocks=100000,strip_min=512,strip_layer_min=384" "strip_min=512,strip_layer_min=384"; do
  PGCN_TUNING="$t" python bench.py --emulate-rank 0/8 --graph --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rank_0_8_$t.json 2>/dev/null
  python -c "
import json; r=json.load(open('$out/bench_rank_0_8_$t.json')); print('$t', 'ms/epoch %.3f'%r['ms_per_step'], 'replay', r.get('graph_replay',{}).get('ms_per_step'), 'loc %.4f'%r['roofline']['avg_launch_ms'], [round(h['avg_launch_ms'],4) for h in r['halo_groups']])"
done
