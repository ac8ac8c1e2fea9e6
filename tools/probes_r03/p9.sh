#!/bin/bash
# r03 batch 9: adaptive task chunk on the rank shapes; new GAT shard test; N = 1 check
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p9; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_gpu.py -m gpu -x -q -k "shard_rank_of_four" > $out/pytest_gat.txt 2>&1; tail -3 $out/pytest_gat.txt
timeout 900 python -m pytest tests/test_hip_gpu.py -m gpu -x -q > $out/pytest_hip.txt 2>&1; tail -2 $out/pytest_hip.txt
run() { tag=$1; rp=$2; tun=$3
  PGCN_TUNING="$tun" python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/b_$tag.json 2> $out/b_$tag.err
  python - <<PY
import json
try:
    r=json.load(open("$out/b_$tag.json")); h=r.get("halo_groups") or []
    print("%-22s ms/epoch %.3f  A_loc %.3f ms bwd %.3f %s | halo %s" % ("$tag", r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"].get("avg_launch_ms_backward_AT", 0), {k: round(v) for k, v in (r["roofline"].get("split_us") or {}).items()}, ["%.3f" % x["avg_launch_ms"] for x in h]))
except Exception as e: print("$tag failed", e)
PY
}
for rp in 0/8 0/4 0/2; do t=$(echo $rp | tr '/' '_')
  run adapt_$t $rp ""
  run fixed_$t $rp "spmm_adaptive_chunk=0"
done
run adapt_cmin_0_8 0/8 "core_min_nnz=2000000"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; python -c "
import json; r=json.load(open('$out/bench_n1.json')); print('N=1 ms/epoch', r['ms_per_step'], 'spmm', r['roofline']['avg_launch_ms'], r['roofline']['split_us'])"
