#!/bin/bash
# strips (half-footprint kernel, side stream) overlapped with the gather kernel (main stream)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p35; rm -rf $out; mkdir -p $out
timeout 300 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip" > $out/tests0.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests0.txt | tail -3
PGCN_CORE_OVERLAP=2 timeout 300 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "strip or spmm" > $out/tests2.txt 2>&1; grep -E "passed|failed|error|^E  " $out/tests2.txt | tail -3
run() { tag=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>$out/$tag.err > $out/$tag.json
  python -c "
import json;r=json.load(open('$out/$tag.json'));print('$tag',r['ms_per_step'],r['roofline']['avg_launch_ms'],r['roofline'].get('avg_launch_ms_backward_AT'))"
}
run base X=1
run ov2 PGCN_CORE_OVERLAP=2
run ov2_pad PGCN_CORE_OVERLAP=2 PGCN_GATHER_LDS_PAD=24000
run ov2_prio PGCN_CORE_OVERLAP=2 PGCN_SIDE_PRIO=1
run ov2_prio_pad PGCN_CORE_OVERLAP=2 PGCN_SIDE_PRIO=1 PGCN_GATHER_LDS_PAD=24000
run ov1 PGCN_CORE_OVERLAP=1
