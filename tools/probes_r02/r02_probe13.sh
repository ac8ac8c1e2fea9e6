#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p13; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gat_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "multi_head or portable or cli or run_ or engine" > $out/tests.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests.txt | tail -12
python tools/gat_probe.py > $out/gat_probe.json 2>$out/gat_probe.err; python -c "
import json;r=json.load(open('$out/gat_probe.json'));print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})"
PGCN_GAT_MULTIHEAD=0 python bench.py --workload reddit-gat --steps 5 --warmup 2 2>$out/bench_gat_k.err > $out/bench_gat_perhead.json; python -c "
import json;r=json.load(open('$out/bench_gat_perhead.json'));print('per-head',r['ms_per_step'],r['ms_per_layer_fwd_bwd'],r['roofline'])"
python bench.py --workload reddit-gat --steps 5 --warmup 2 2>$out/bench_gat.err > $out/bench_gat.json; python -c "
import json;r=json.load(open('$out/bench_gat.json'));print('multi-head',r['ms_per_step'],r['ms_per_layer_fwd_bwd'],r['roofline'])"
