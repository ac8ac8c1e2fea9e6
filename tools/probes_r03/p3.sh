#!/bin/bash
# r03 batch 3: effective per-XCD L2 capacity for row gathers -- uniform hot sets of K rows (K/8 rows per XCD slice),
# 128-B rows (f = 32) and 512-B rows (f = 128): TCC hit rate + kernel time vs working-set size
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p3; rm -rf $out; mkdir -p $out
for f in 32 128; do
  for K in 8192 16384 32768 65536 131072 232965; do
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "spmm_tasks" --output-format csv -d $out/pmc_f${f}_K$K -- python tools/spmm_probe.py --f $f --workload uniform:$K --once s8c1024 > $out/log_f${f}_K$K.txt 2>&1
    python tools/pmc_summary.py $out/pmc_f${f}_K$K spmm_tasks > $out/sum_f${f}_K$K.txt
    echo "f=$f K=$K WS/XCD=$((K/8*f*4/1024)) KB: $(grep -E 'TCC_HIT|TCC_MISS' $out/sum_f${f}_K$K.txt | tr -s ' ' | tr '\n' ' ') $(grep 'mean=.*us' $out/sum_f${f}_K$K.txt | sed 's/.*| n=/n=/')"
  done
done
