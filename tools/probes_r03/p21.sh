#!/bin/bash
# the fused kernel with its reduction pipelined under the next batch's gathers: kernel tests, pieces, GAT bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p21; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_gpu.py -q -m gpu -x -k "fused or recomputed or multi_head or heads or layers_multi_rank" > $out/pytest_gat.txt 2>&1; tail -4 $out/pytest_gat.txt
timeout 300 python tools/gat_probe.py > $out/gat_probe.log 2>&1; cp gpurun_out/gat_probe_standard.json $out/ 2>/dev/null; grep -E "forward_ms|backward_ms|spmm_heads_kernel_ms|heads_recompute_T_ms|heads_grad_fused_ms|row_sums" $out/gat_probe.log
timeout 300 python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat.json 2> $out/bench_gat.err; python - <<PY
import json
r=json.load(open("$out/bench_gat.json")); print("GAT ms/epoch", r["ms_per_step"], r["roofline"].get("avg_launch_ms"))
PY
