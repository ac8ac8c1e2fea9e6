#!/bin/bash
# the edge gradient fused into the transposed product (pgcn_spmm_heads_grad_f32): tests, pieces, the GAT bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p19; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_gat_gpu.py -x -q -m gpu > $out/pytest_gat.txt 2>&1; tail -5 $out/pytest_gat.txt
timeout 300 python tools/gat_probe.py > $out/gat_probe.log 2>&1; cp gpurun_out/gat_probe_standard.json $out/ 2>/dev/null; tail -32 $out/gat_probe.log
timeout 300 python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat.json 2> $out/bench_gat.err; python - <<PY
import json
r=json.load(open("$out/bench_gat.json")); print("GAT ms/epoch", r["ms_per_step"], r["roofline"].get("kernel","")[:80], r["roofline"].get("avg_launch_ms"))
PY
PGCN_TUNING=gat_fused_grad=0 timeout 300 python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat_unfused.json 2>/dev/null; python - <<PY
import json
r=json.load(open("$out/bench_gat_unfused.json")); print("GAT unfused ms/epoch", r["ms_per_step"])
PY
