#!/bin/bash
# kernel-trace of one SpMM variant (per-kernel durations):  trace_probe.sh <variant> [extra probe args]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
v=$1; shift
rm -rf gpurun_out/trace_$v; mkdir -p gpurun_out/trace_$v
rocprofv3 --kernel-trace --kernel-include-regex "spmm" --output-format csv -d gpurun_out/trace_$v -- python tools/spmm_probe.py --once $v "$@" > gpurun_out/trace_$v/log.txt 2>&1
python tools/pmc_summary.py gpurun_out/trace_$v spmm
