#!/bin/bash
# r03 batch 7: column-sweep feasibility (merged rows), small-block single-kernel path on the rank shapes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p7; rm -rf $out; mkdir -p $out
python tools/sweep_probe.py --merge 1,2,4,8,16,32,64,128 > $out/sweep.txt 2>&1; cat $out/sweep.txt | grep -v Warn
python tools/sweep_probe.py --merge 8,32 --interleave 1 > $out/sweep_il.txt 2>&1; grep merge $out/sweep_il.txt
for m in 1 8 32; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "spmm_tasks" --output-format csv -d $out/pmc_m$m -- python tools/sweep_probe.py --once $m > $out/pmc_m$m.log 2>&1
  python tools/pmc_summary.py $out/pmc_m$m spmm_tasks > $out/pmc_sum_m$m.txt
  echo "m=$m $(grep -E 'TCC_HIT|TCC_MISS' $out/pmc_sum_m$m.txt | tr -s ' ' | tr '\n' ' ') $(grep 'mean=.*us' $out/pmc_sum_m$m.txt | sed 's/.*| n=/n=/')"
done
run() { tag=$1; rp=$2; tun=$3
  PGCN_TUNING="$tun" python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/b_$tag.json 2> $out/b_$tag.err
  python - <<PY
import json
try:
    r=json.load(open("$out/b_$tag.json")); h=r.get("halo_groups") or []
    print("%-22s ms/epoch %.3f  A_loc %.3f ms %s | halo %s" % ("$tag", r["ms_per_step"], r["roofline"]["avg_launch_ms"], {k: round(v) for k, v in (r["roofline"].get("split_us") or {}).items()}, ["%.3f" % x["avg_launch_ms"] for x in h]))
except Exception as e: print("$tag failed", e)
PY
}
run base_0_8 0/8 ""
run cmin2m_0_8 0/8 "core_min_nnz=2000000"
run cmin2m_r1_0_8 0/8 "core_min_nnz=2000000,exchange_rounds=1"
