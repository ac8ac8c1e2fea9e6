# Records for the GEMM shapes of the packed GAT projection (n x 256 x 264 and its two backward products) in the shipped TunableOp file:
# PyTorch's TunableOp times them once here (PGCN_TUNING=gemm_tunableop=1), the merged file replaces tunableop/gfx950.csv.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
PKG=scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd
out=gpurun_out/r05_p14; rm -rf $out; mkdir -p $out
cp $PKG/tunableop/gfx950.csv $out/cache.csv
export PGCN_TUNABLEOP_CACHE=$PWD/$out/cache.csv
PGCN_TUNING=gemm_tunableop=1 python bench.py --workload reddit-gat --steps 3 --warmup 2 --no-cpu-baseline > $out/bench_tune.json 2> $out/bench_tune.err
python -c "
import json; r=json.load(open('$out/bench_tune.json')); print('tuning run: setup_s %.1f'%r['setup_s'], 'ms %.3f'%r['ms_per_step'])" || tail -5 $out/bench_tune.err
diff <(sort $PKG/tunableop/gfx950.csv) <(sort $out/cache.csv) | tee $out/new_records.txt
cp $out/cache.csv $out/gfx950_merged.csv
