#!/bin/bash
# r04 batch 21: after the size-dependent thresholds (tuning.strip_big_nnz, dense3_min_blocks): the bench lines again (N = 1, the
# emulated ranks) and the PMC traffic records re-taken on the final sources (the pmc part of tools/final_profile.sh)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=r04; out=gpurun_out/r04_p21; rm -rf $out; mkdir -p $out
pmc() {
  name=$1; key="$2 $3 $4 $5 $6 $7"; shift 7
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo "$set" | tr ' ' '+')
    rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/pmc_$name/$t -- python tools/group_probe.py "$@" > $out/pmc_${name}_$t.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_$name spmm > $out/pmc_summary_$name.txt
  python tools/make_pmc_traffic.py $out/pmc_summary_$name.txt $out/pmc_traffic.json profiles/${tag}_pmc_$name.txt $key
  rm -rf $out/pmc_$name
}
pmc reddit      reddit rmat 1 128 random loc
pmc reddit_r8h0 reddit rmat 0/8 128 random halo0 --emulate-rank 0/8 --block halo0
pmc reddit_r8l  reddit rmat 0/8 128 random loc --emulate-rank 0/8 --block loc
pmc products    products rmat 1 128 random loc --workload products
pmc reddit_sbm  reddit sbm 1 128 random loc --generator sbm
python tools/make_shards.py --workload papers --ranks 8 --only-rank 0 --device cuda --out /tmp/papers > $out/papers_make_shards.txt 2>&1
pmc papers_r8l  papers rmat 0/8 64 block loc   --workload papers --shards /tmp/papers --emulate-rank 0/8 --features 64 --block loc
pmc papers_r8h0 papers rmat 0/8 64 block halo0 --workload papers --shards /tmp/papers --emulate-rank 0/8 --features 64 --block halo0
cp $out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_stdout.log 2> $out/prof_stderr.log
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
python bench.py --workload products --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_products.json 2>/dev/null
python bench.py --generator sbm --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_sbm.json 2>/dev/null
python bench.py --workload mid --steps 10 --warmup 2 > $out/bench_mid.json 2>/dev/null
python bench.py --workload papers --emulate-rank 0/8 --shards /tmp/papers --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_papers_full_rank_0_8.json 2>/dev/null
for rp in 0/8 3/8 7/8 0/4 0/2; do t=$(echo $rp | tr '/' '_')
  python bench.py --emulate-rank $rp --graph --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rank_$t.json 2>/dev/null
done
for f in $out/bench*.json; do python - <<PY
import json
try:
    r=json.load(open("$f")); ro=r.get("roofline") or {}
    print("%-40s ms/step %8.3f replay %s group %.4f ms frac %.4f traffic %s halo %s setup %.1f" % ("$(basename $f)", r["ms_per_step"], (r.get("graph_replay") or {}).get("ms_per_step"), ro.get("avg_launch_ms", 0), ro.get("frac", 0), ro.get("traffic"), [round(h["avg_launch_ms"], 3) for h in (r.get("halo_groups") or [])], r.get("setup_s", 0)))
except Exception as e: print("$(basename $f)", "FAILED", e)
PY
done
