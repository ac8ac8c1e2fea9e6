#!/bin/bash
# r03 batch 15: 8-wave MFMA-tile kernel (two waves per row block split the feature blocks) vs the 4-wave shape
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p15; rm -rf $out; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_gpu.py -m gpu -x -q -k "dense or mfma or spmm or inf" > $out/pytest.txt 2>&1; grep -E "passed|failed" $out/pytest.txt | tail -2
python tools/spmm_probe.py --variants s8c1024k_p64 --libs w4,w8 --split --check --rounds 10 > $out/probe.txt 2>&1; grep -E "median|split|diff" $out/probe.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "dense" --output-format csv -d $out/pmc_mfma -- python tools/group_probe.py > $out/pmc_mfma.log 2>&1
python tools/pmc_summary.py $out/pmc_mfma dense
for i in 1 2; do python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $out/b_$i.json 2>/dev/null; python -c "
import json; r=json.load(open('$out/b_$i.json')); print('w8 bench ms/epoch %.3f spmm %.4f %s' % (r['ms_per_step'], r['roofline']['avg_launch_ms'], {k: round(v) for k, v in r['roofline']['split_us'].items()}))"; done
