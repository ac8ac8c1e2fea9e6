#!/bin/bash
# forward product with recomputed weights + second accumulator (no alpha planes, no de): the GAT file, pieces, bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p22; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_gpu.py -q -m gpu > $out/pytest_gat.txt 2>&1; tail -12 $out/pytest_gat.txt
timeout 300 python tools/gat_probe.py > $out/gat_probe.log 2>&1; cp gpurun_out/gat_probe_standard.json $out/ 2>/dev/null; grep -E "forward_ms|backward_ms|softmax|spmm_heads_kernel_ms|heads_recompute_T_ms|heads_grad_fused_ms|forward2|row_sums" $out/gat_probe.log
timeout 300 python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat.json 2> $out/bench_gat.err; python - <<PY
import json
r=json.load(open("$out/bench_gat.json")); print("GAT ms/epoch", r["ms_per_step"], r["roofline"].get("avg_launch_ms"), "loss", r["loss"])
PY
PGCN_TUNING=gat_fused_forward=0 timeout 300 python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat_nofwd2.json 2>/dev/null; python - <<PY
import json
r=json.load(open("$out/bench_gat_nofwd2.json")); print("GAT (planes forward) ms/epoch", r["ms_per_step"], "loss", r["loss"])
PY
