#!/bin/bash
# last measurements of the round: the headline step as a HIP-graph replay, config 3 at FULL size on rank 0 of 8
# (community-block vector vs random), PMC traffic of the GAT backward pass
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p24; rm -rf $out; mkdir -p $out
python bench.py --graph --no-cpu-baseline > $out/bench_graph.json 2> $out/bench_graph.err
python bench.py --workload products --generator sbm --partvec tests/golden/partvec/products-sbm.A.mtx.8.cb.gz --emulate-rank 0/8 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_products_sbm_cb_rank_0_8.json 2>/dev/null
python bench.py --workload products --generator sbm --emulate-rank 0/8 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_products_sbm_rp_rank_0_8.json 2>/dev/null
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "spmm_heads" --output-format csv -d $out/pmc_gat/$set -- python bench.py --workload reddit-gat --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/pmc_gat_$set.log 2>&1
done
python tools/pmc_summary.py $out/pmc_gat spmm_heads > $out/pmc_summary_gat.txt
cp profiles/pmc_traffic.json $out/pmc_traffic.json
python tools/make_pmc_traffic.py $out/pmc_summary_gat.txt $out/pmc_traffic.json profiles/r03_pmc_gat.txt reddit-gat rmat 1 256 random gat_grad "spmm_heads_kernel<4, true, true, false>"
rm -rf $out/pmc_gat
cat $out/pmc_summary_gat.txt
for f in $out/bench*.json; do python - <<PY
import json
try:
    r=json.load(open("$f")); ro=r.get("roofline") or {}
    print("%-44s ms/step %8.3f group %.3f halo %s graph %s shape %s" % ("$(basename $f)", r["ms_per_step"], ro.get("avg_launch_ms", 0), [round(h["avg_launch_ms"], 3) for h in (r.get("halo_groups") or [])], (r.get("graph_replay") or {}).get("ms_per_step"), r["config"].get("rank_shape")))
except Exception as e: print("$(basename $f)", "FAILED", e)
PY
done
