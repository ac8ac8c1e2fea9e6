#!/bin/bash
# Round evidence in one gpurun call: GPU tests, smoke, default bench, rocprofv3 kernel stats of the bench command, PMC
# traffic passes per launch group (tools/group_probe.py: the block exactly as bench.py builds it), the other workloads
# (products, SBM, mid, GAT), the shard shapes (--emulate-rank) and one rank of the papers100M shape from its shard.
# usage: [FINAL_TESTS_K=expr] bash tools/final_profile.sh r04 [hp-partvec workload]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=${1:-r06}
HP=${2:-tests/golden/partvec/products4-sbm.A.mtx.8.hp.gz}; W3=${3:-products4}
out=gpurun_out/final_$tag; rm -rf $out; mkdir -p $out
# (FINAL_TESTS_K="expr": only the GPU tests matching it -- when the full suite already ran in its own call)
if [ -n "${FINAL_TESTS_K:-}" ]; then timeout 2400 python -m pytest tests -m gpu -q -k "$FINAL_TESTS_K" > $out/pytest_gpu_full.txt 2>&1
else timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu_full.txt 2>&1; fi
grep -E "passed|failed|error" $out/pytest_gpu_full.txt | tail -3 | tee $out/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.txt
pmc() {  # name, then the record key: workload generator ranks f partvec block, then group_probe arguments
  name=$1; key="$2 $3 $4 $5 $6 $7"; shift 7
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo "$set" | tr ' ' '+')
    rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm" --output-format csv -d $out/pmc_$name/$t -- python tools/group_probe.py "$@" > $out/pmc_${name}_$t.log 2>&1
  done
  python tools/pmc_summary.py $out/pmc_$name spmm > $out/pmc_summary_$name.txt
  python tools/make_pmc_traffic.py $out/pmc_summary_$name.txt $out/pmc_traffic.json profiles/${tag}_pmc_$name.txt $key
  rm -rf $out/pmc_$name
}
if [ -z "${FINAL_SKIP_PMC:-}" ]; then      # (FINAL_SKIP_PMC=1: the stamped sources did not change since the last passes -- keep profiles/pmc_traffic.json)
pmc reddit      reddit rmat 1 128 random loc
pmc reddit_r8h0 reddit rmat 0/8 128 random halo0 --emulate-rank 0/8 --block halo0
pmc reddit_r8l  reddit rmat 0/8 128 random loc --emulate-rank 0/8 --block loc
pmc products    products rmat 1 128 random loc --workload products
pmc reddit_sbm  reddit sbm 1 128 random loc --generator sbm
# BASELINE config 4 on one rank at full size: shard + degree vector made here (6 s on the GPU), then its local block / first halo group
python tools/make_shards.py --workload papers --ranks 8 --only-rank 0 --device cuda --out /tmp/papers > $out/papers_make_shards.txt 2>&1
pmc papers_r8l  papers rmat 0/8 64 block loc   --workload papers --shards /tmp/papers --emulate-rank 0/8 --features 64 --block loc
pmc papers_r8h0 papers rmat 0/8 64 block halo0 --workload papers --shards /tmp/papers --emulate-rank 0/8 --features 64 --block halo0
# BASELINE config 5: the dominant pass of the GAT epoch (the fused transposed product + edge gradient: the gather kernel over the entries outside the
# dense blocks + the block kernel, r06), counters over the bench command itself
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$set" | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm_heads|gat_blocks" --output-format csv -d $out/pmc_gat/$t -- python bench.py --workload reddit-gat --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/pmc_gat_$t.log 2>&1
done
python tools/pmc_summary.py $out/pmc_gat "spmm_heads|gat_blocks" > $out/pmc_summary_gat.txt
python tools/make_pmc_traffic.py $out/pmc_summary_gat.txt $out/pmc_traffic.json profiles/${tag}_pmc_gat.txt reddit-gat rmat 1 256 random gat_grad "spmm_heads_kernel<4, true, true, false>|gat_blocks_kernel<true"
rm -rf $out/pmc_gat
# ... and of rank 0 of 4 of the same line.  N > 1: the transposed structure runs as its halo rows and its local rows (two
# launches of the same kernel per layer); the record is their mean per launch
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$set" | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm_heads|gat_blocks" --output-format csv -d $out/pmc_gat_r4/$t -- python bench.py --workload reddit-gat --emulate-rank 0/4 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/pmc_gat_r4_$t.log 2>&1
done
python tools/pmc_summary.py $out/pmc_gat_r4 "spmm_heads|gat_blocks" > $out/pmc_summary_gat_r4.txt
python tools/make_pmc_traffic.py $out/pmc_summary_gat_r4.txt $out/pmc_traffic.json profiles/${tag}_pmc_gat_r4.txt reddit-gat rmat 0/4 256 random gat_grad "spmm_heads_kernel<4, true, true, false>|gat_blocks_kernel<true"
rm -rf $out/pmc_gat_r4
cp $out/pmc_traffic.json profiles/pmc_traffic.json      # (bench.py reads it from there for the lines below)
else python tools/make_shards.py --workload papers --ranks 8 --only-rank 0 --device cuda --out /tmp/papers > $out/papers_make_shards.txt 2>&1; fi
python bench.py > $out/bench.json 2> $out/bench.err; tail -c 1200 $out/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_stdout.log 2> $out/prof_stderr.log
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
python bench.py --workload products --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_products.json 2>/dev/null
python bench.py --generator sbm --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_sbm.json 2>/dev/null
python bench.py --workload mid --steps 10 --warmup 2 > $out/bench_mid.json 2>/dev/null
python bench.py --workload reddit-gat --steps 5 --warmup 2 > $out/bench_gat.json 2>/dev/null
PGCN_TUNING=gat_blocks=0 python bench.py --workload reddit-gat --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat_noblocks.json 2>/dev/null   # (the same line with every entry in the gather kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_gat -o gat -- python bench.py --workload reddit-gat --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/prof_gat_stdout.log 2> $out/prof_gat_stderr.log
rm -f $out/prof_gat/*kernel_trace.csv $out/prof_gat/*/*kernel_trace.csv
python bench.py --workload papers --emulate-rank 0/8 --shards /tmp/papers --partvec block --features 64 --layers 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_papers_full_rank_0_8.json 2>/dev/null
python tools/shard_rank_check.py --shards /tmp/papers --rank 0 --ranks 8 --features 64 > $out/papers_full_rank_0_8_check.json 2>/dev/null
for rp in 0/8 3/8 7/8 0/4 0/2; do t=$(echo $rp | tr '/' '_')
  python bench.py --emulate-rank $rp --graph --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_rank_$t.json 2>/dev/null
done
python bench.py --workload reddit-gat --emulate-rank 0/4 --steps 5 --warmup 2 > $out/bench_gat_rank_0_4.json 2>/dev/null
python bench.py --workload reddit-gat --emulate-rank 0/4 --pace-exchange 153 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_gat_rank_0_4_paced.json 2>/dev/null   # (exchange priced at the xGMI link rate)
if [ -f "$HP" ]; then
  for r in 0 3; do
    python bench.py --workload $W3 --generator sbm --partvec $HP --emulate-rank $r/8 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_${W3}_sbm_hp_rank_${r}_8.json 2>/dev/null
    python bench.py --workload $W3 --generator sbm --emulate-rank $r/8 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_${W3}_sbm_rp_rank_${r}_8.json 2>/dev/null
  done
fi
for f in $out/bench*.json; do python - <<PY
import json
try:
    r=json.load(open("$f")); ro=r.get("roofline") or {}
    print("%-48s ms/step %8.3f  group %.3f ms frac %.4f traffic %s  halo %s" % ("$(basename $f)", r["ms_per_step"], ro.get("avg_launch_ms", 0), ro.get("frac", 0), ro.get("traffic"), [round(h["avg_launch_ms"], 3) for h in (r.get("halo_groups") or [])]))
except Exception as e: print("$(basename $f)", "FAILED", e)
PY
done

# the evidence the round's documents cite, copied where it is tracked
cp $out/bench.json profiles/${tag}_bench_stdout.json 2>/dev/null
cp $out/pytest_gpu_full.txt profiles/${tag}_pytest_gpu.txt 2>/dev/null
f=$(ls $out/prof/*/*kernel_stats.csv $out/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${tag}_bench_kernel_stats.csv
f=$(ls $out/prof_gat/*/*kernel_stats.csv $out/prof_gat/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/${tag}_bench_gat_kernel_stats.csv
for w in products sbm mid gat gat_noblocks; do cp $out/bench_$w.json profiles/${tag}_bench_${w}_stdout.json 2>/dev/null; done
for f in $out/bench_rank_*.json $out/bench_gat_rank_0_4.json $out/bench_papers_full_rank_*.json $out/bench_${W3}_sbm_*_rank_*.json $out/papers_full_rank_0_8_check.json; do
  [ -f "$f" ] && cp "$f" profiles/${tag}_$(basename $f)
done
[ -f gpurun_out/parity_observed.jsonl ] && cp gpurun_out/parity_observed.jsonl profiles/${tag}_parity_observed.jsonl
ls profiles | grep "^${tag}_" | wc -l
