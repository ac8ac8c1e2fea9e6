#!/bin/bash
# r03 batch 16: GAT backward -- edge gradient stored entry-major (one 16-byte gather per entry for ds2), row statistics of
# all heads in one pass of the transposed walk
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p16; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gat_gpu.py tests/test_capi_symbols.py -m gpu -x -q > $out/pytest_gat.txt 2>&1; grep -E "passed|failed" $out/pytest_gat.txt | tail -2
python tools/gat_probe.py > $out/gat_probe.json 2> $out/gat_probe.err; python -c "
import json; r=json.load(open('gpurun_out/gat_probe_standard.json')); print({k: round(v, 2) for k, v in r.items() if k.endswith('_ms')})"
python bench.py --workload reddit-gat --steps 5 --warmup 2 > $out/bench_gat.json 2>/dev/null; python -c "
import json; r=json.load(open('$out/bench_gat.json')); print('GAT ms/epoch %.2f per layer %.2f' % (r['ms_per_step'], r['ms_per_layer_fwd_bwd']))"
