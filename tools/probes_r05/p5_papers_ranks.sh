#!/bin/bash
# BASELINE config 4 at FULL size: ranks 3 and 7 of 8 of the papers100M shape beside r04's rank 0 (load balance of the emulated ranks):
# shard + degree vector on the GPU (tools/make_shards.py --only-rank), the rank's training step with a no-op exchange.
# gpurun --timeout 1200 -- 'bash tools/probes_r05/p5_papers_ranks.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p5; rm -rf $out; mkdir -p $out
for r in ${RANKS:-3 7}; do
  ( time python tools/make_shards.py --workload papers --ranks 8 --only-rank $r --device cuda --out /tmp/papers ) > $out/make_shards_$r.txt 2>&1; tail -4 $out/make_shards_$r.txt
  cp /tmp/papers.meta.json $out/meta_$r.json
  timeout 500 python bench.py --emulate-rank $r/8 --shards /tmp/papers --features 64 --layers 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_papers_full_rank_${r}_8.json 2> $out/bench_$r.err
  python -c "
import json; r=json.load(open('$out/bench_papers_full_rank_${r}_8.json')); print('rank $r ms/epoch %.2f'%r['ms_per_step'], r['config']['rank_shape'], 'A_loc group %.3f ms'%r['roofline']['avg_launch_ms'], [(h['round'], round(h['avg_launch_ms'],3)) for h in r.get('halo_groups',[])])" || tail -5 $out/bench_$r.err
  rm -f /tmp/papers.*
done
