#!/bin/bash
# r04 batch 13: BASELINE config 4 at FULL size on one rank -- the papers100M shape (n = 111 059 956, 1.6 G entries), rank 0 of
# 8 under contiguous blocks (the generator permutes the vertex ids: a block vector is a random-like partition of this graph),
# f = 64, 2 layers.  Shard + degree vector on the GPU (tools/make_shards.py --only-rank), full-size parity against float64
# (tools/shard_rank_check.py), the rank's training step with a no-op exchange (bench.py --emulate-rank 0/8 --shards).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r04_p13; rm -rf $out; mkdir -p $out
SCALE=${1:-1.0}
export PGCN_TUNABLEOP_CACHE=$PWD/$out/tunableop_cache.csv      # (GEMM choices of these shapes: merged into the shipped file afterwards)
( time python tools/make_shards.py --workload papers --ranks 8 --only-rank 0 --device cuda --scale $SCALE --out /tmp/papers ) > $out/make_shards.txt 2>&1; tail -4 $out/make_shards.txt
ls -la /tmp/papers* >> $out/make_shards.txt; cp /tmp/papers.meta.json $out/
timeout 900 python tools/shard_rank_check.py --shards /tmp/papers --rank 0 --ranks 8 --features 64 > $out/check.json 2> $out/check.err; tail -c 1500 $out/check.json; tail -3 $out/check.err
timeout 900 python bench.py --emulate-rank 0/8 --shards /tmp/papers --features 64 --layers 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_papers_rank_0_8.json 2> $out/bench.err; tail -c 1500 $out/bench_papers_rank_0_8.json; tail -3 $out/bench.err
