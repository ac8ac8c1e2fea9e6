#!/bin/bash
# PMC passes for the SpMM kernel (separate passes: TCC has 4 slots, FETCH_SIZE takes 3).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
for v in "$@"; do
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    tag=$(echo "$set" | tr ' ' '+' | cut -c1-40)
    rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d gpurun_out/pmc/$v/$tag -- python tools/spmm_probe.py --once $v > gpurun_out/pmc/$v.$tag.log 2>&1 || echo "FAILED $v $set"
  done
  python tools/pmc_summary.py gpurun_out/pmc/$v spmm_tasks
done
