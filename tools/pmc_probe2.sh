#!/bin/bash
# reduced PMC passes:  pmc_probe2.sh <variant>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
v=$1
rm -rf gpurun_out/pmc/$v; mkdir -p gpurun_out/pmc/$v
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo "$set" | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d gpurun_out/pmc/$v/$tag -- python tools/spmm_probe.py --once $v > gpurun_out/pmc/$v.$tag.log 2>&1 || echo "FAILED $v $set"
done
python tools/pmc_summary.py gpurun_out/pmc/$v spmm | grep -v kernel_trace
