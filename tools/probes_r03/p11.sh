#!/bin/bash
# r03 batch 11: rows finalised by their gather task -- bit identity tests, N = 1 bench A/B, shard shapes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p11; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "not products_sbm" > $out/pytest.txt 2>&1; grep -E "passed|failed" $out/pytest.txt | tail -2
for fin in 1 0 1 0; do
  PGCN_TUNING="finalise_rows=$fin" python bench.py --steps 15 --warmup 3 --no-cpu-baseline > $out/bench_fin$fin.json 2> $out/bench_fin$fin.err
  python -c "
import json; r=json.load(open('$out/bench_fin$fin.json')); print('finalise=$fin N=1 ms/epoch %.3f' % r['ms_per_step'], 'spmm %.4f' % r['roofline']['avg_launch_ms'], 'bwd %.4f' % r['roofline']['avg_launch_ms_backward_AT'], {k: round(v) for k, v in r['roofline']['split_us'].items()})"
done
for rp in 0/8 0/4; do t=$(echo $rp | tr '/' '_')
  for fin in 1 0; do
  PGCN_TUNING="finalise_rows=$fin" python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/b_${t}_$fin.json 2>/dev/null
  python -c "
import json; r=json.load(open('$out/b_${t}_$fin.json')); print('finalise=$fin rank $rp ms/epoch %.3f' % r['ms_per_step'], 'A_loc %.3f' % r['roofline']['avg_launch_ms'], [round(h['avg_launch_ms'],3) for h in r['halo_groups']])"
  done
done
