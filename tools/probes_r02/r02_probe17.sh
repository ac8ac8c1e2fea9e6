#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p17; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gat_gpu.py -m gpu -x -q > $out/tests.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests.txt | tail -6
python tools/gat_probe.py > $out/gat_probe.json 2>$out/gat_probe.err; python -c "
import json;r=json.load(open('$out/gat_probe.json'));print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})"
