#!/bin/bash
# Regenerates <package>/tunableop/gfx950.csv -- the GEMM choices that ship with the package (PGCN.tune_dense_gemms) -- on an
# MI355X: runs the set-up of the benchmark workloads with an empty per-machine cache, so that every n x f x f shape of
# theirs is timed, and copies the cache.   usage: gpurun --timeout 600 -- 'bash tools/make_tunableop.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
PKG=scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd
out=gpurun_out/tunableop; rm -rf $out; mkdir -p $out
export PGCN_TUNABLEOP_CACHE=$PWD/$out/gfx950.csv
export PGCN_TUNING=gemm_tunableop=1     # records are MADE by PyTorch's TunableOp (the default only replays them)
mv $PKG/tunableop/gfx950.csv $out/shipped_before.csv 2>/dev/null
for w in "--workload reddit" "--workload products" "--workload reddit-gat" "--workload mid"; do
  python bench.py $w --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_$(echo $w | tr -d ' -').json 2> $out/bench_$(echo $w | tr -d ' -').err
  python -c "
import json,sys; r=json.load(open('$out/bench_$(echo $w | tr -d ' -').json')); print('$w', 'setup_s %.1f'%r['setup_s'], 'ms %.3f'%r['ms_per_step'])"
done
wc -l $out/gfx950.csv; mkdir -p $PKG/tunableop; cp $out/gfx950.csv $PKG/tunableop/gfx950.csv
# a second process with the shipped file and NO cache, default mode (records replayed by solution index): nothing is timed
PGCN_TUNING= PGCN_TUNABLEOP_CACHE=/tmp/none.csv python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_again.json 2> $out/bench_again.err
python -c "
import json; r=json.load(open('$out/bench_again.json')); print('again: setup_s %.1f'%r['setup_s'], 'ms %.3f'%r['ms_per_step'])"; ls -la /tmp/none.csv 2>&1 | tail -1
