#!/bin/bash
# r03 batch 5: strip tiles on the shard shapes with fewer, longer pieces; one exchange round
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p5; rm -rf $out; mkdir -p $out
run() {  # tag, rank spec, env...
  tag=$1; rp=$2; shift 2
  env "$@" python bench.py --emulate-rank $rp --steps 10 --warmup 2 --no-cpu-baseline > $out/b_$tag.json 2> $out/b_$tag.err
  python - <<PY
import json
try:
    r=json.load(open("$out/b_$tag.json"))
    h=r.get("halo_groups") or []
    print("%-28s ms/epoch %.3f  A_loc %.3f ms %s | halo %s" % ("$tag", r["ms_per_step"], r["roofline"]["avg_launch_ms"], {k: round(v) for k, v in (r["roofline"].get("split_us") or {}).items()}, ["%.3f" % x["avg_launch_ms"] for x in h]))
except Exception as e:
    print("$tag failed", e)
PY
}
for rp in 0/8 0/4; do
  t=$(echo $rp | tr '/' '_')
  run base_$t $rp PGCN_X=0
  run s128_$t $rp PGCN_STRIP_MIN_RECORDS=0 PGCN_STRIP_PIECES=128
  run s256_$t $rp PGCN_STRIP_MIN_RECORDS=0 PGCN_STRIP_PIECES=256
  run s512_$t $rp PGCN_STRIP_MIN_RECORDS=0 PGCN_STRIP_PIECES=512
  run r1_$t $rp PGCN_EXCHANGE_ROUNDS=1
  run r1s256_$t $rp PGCN_EXCHANGE_ROUNDS=1 PGCN_STRIP_MIN_RECORDS=0 PGCN_STRIP_PIECES=256
done
