#!/bin/bash
# r03 batch 4: the shard shapes (bench.py --emulate-rank r/P): per-rank epoch, A_loc / A_halo launch groups, kernel stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p4; rm -rf $out; mkdir -p $out
for rp in 0/8 0/4 0/2; do
  t=$(echo $rp | tr '/' '_')
  python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_$t.json 2> $out/bench_$t.err
  python - <<PY
import json
r=json.load(open("$out/bench_$t.json"))
print("$rp ms/epoch %.3f"%r["ms_per_step"], "A_loc ms", r["roofline"]["avg_launch_ms"], "split", r["roofline"].get("split_us"), "ps/entry", r["roofline"].get("ps_per_entry"))
for h in r.get("halo_groups") or []: print("   halo round", h["round"], "nnz", h["nnz"], "ms", h["avg_launch_ms"], "ps/entry", h["ps_per_entry"], "frac", h["frac"])
print("   shape", r["config"]["rank_shape"])
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o r8 -- python bench.py --emulate-rank 0/8 --steps 10 --warmup 2 --no-cpu-baseline > $out/prof_stdout.log 2> $out/prof_stderr.log
f=$(ls $out/prof/*kernel_stats.csv $out/prof/*/*kernel_stats.csv 2>/dev/null | head -1); head -30 $f | cut -c1-200
rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv
