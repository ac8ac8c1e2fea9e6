#!/bin/bash
# r03 batch 2: is the gather part's L2 hit rate a LAYOUT problem?  compact rows (f = 32 / 64 operands), range slices
# ("d:" = degree order dealt into 8 contiguous column ranges) with and without feature passes, unsliced leftovers
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p2; rm -rf $out; mkdir -p $out
for f in 32 64; do
  python tools/spmm_probe.py --f $f --variants s8c1024k,d:s8c1024k --split --rounds 5 > $out/probe_f$f.txt 2>&1
  echo "f=$f"; grep -E "median|split" $out/probe_f$f.txt
done
python tools/spmm_probe.py --variants s8c1024k_p64,d:s8c1024k,d:s8c1024k_p64s,d:s8c1024k_p32s,s1c1024k,s1c1024xk --split --rounds 5 > $out/probe_f128.txt 2>&1
echo "f=128"; grep -E "median|split" $out/probe_f128.txt
pmc() {  # tag, f, variant
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo "$set" | tr ' ' '+')
    rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm_tasks" --output-format csv -d $out/pmc_$1/$t -- python tools/spmm_probe.py --f $2 --once $3 > $out/pmc_$1_$t.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_$1 spmm_tasks > $out/pmc_summary_$1.txt; echo "PMC $1"; grep -v "^FETCH.*trace\|^TCC.*trace" $out/pmc_summary_$1.txt | grep -E "FETCH_SIZE |TCC_|mean=.*us" | head -8
}
pmc f32_s8 32 s8c1024k
pmc f32_d8 32 d:s8c1024k
pmc f128_d8p32s 128 d:s8c1024k_p32s
pmc f128_d8 128 d:s8c1024k
pmc f128_s1 128 s1c1024k
