#!/bin/bash
# r03 batch 10: the new GPU tests (configs 4 / 5, emulated rank, GAT shard), shard shapes with the final small-block rules
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p10; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q -k "papers_shape or gat_two_ranks or emulated_rank or shard_rank_of_four" > $out/pytest_new.txt 2>&1; tail -4 $out/pytest_new.txt | cut -c1-300
run() { tag=$1; rp=$2; tun=$3
  PGCN_TUNING="$tun" python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/b_$tag.json 2> $out/b_$tag.err
  python - <<PY
import json
try:
    r=json.load(open("$out/b_$tag.json")); h=r.get("halo_groups") or []
    print("%-22s ms/epoch %.3f  A_loc %.3f ms bwd %.3f %s | halo %s" % ("$tag", r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"].get("avg_launch_ms_backward_AT", 0), {k: round(v) for k, v in (r["roofline"].get("split_us") or {}).items()}, ["%.3f" % x["avg_launch_ms"] for x in h]))
except Exception as e: print("$tag failed", e)
PY
}
