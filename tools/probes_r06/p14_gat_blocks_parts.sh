# r06: what bounds the block kernel -- timing-only variants (PGCN_GATB_PROBES build): 1 = no MFMAs, 2 = no weight slices, 3 = neither.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p14; rm -rf $out; mkdir -p $out
cp scalable*/lib/libpgcn_hip.so $out/libpgcn_hip.so.keep
PGCN_EXTRA_FLAGS=-DPGCN_GATB_PROBES bash scalable*/csrc/build.sh > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
for p in 0 1 2 3; do
  PGCN_GATB_PROBE=$p PGCN_TUNING="gat_block_tau=0.06" timeout 600 python bench.py --workload reddit-gat --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_$p.json 2> $out/bench_$p.err
  python - $out/bench_$p.json $p <<'PY' || tail -5 $out/bench_$p.err
import json, sys
r = json.load(open(sys.argv[1]))
print('probe', sys.argv[2], 'ms/epoch %.3f' % r['ms_per_step'], {k: (round(v, 3) if v else v) for k, v in r['roofline']['pass_split_ms'].items()})
PY
done
cp $out/libpgcn_hip.so.keep scalable*/lib/libpgcn_hip.so
