# r06: one rank of an 8-way run -- the block grid of the halo blocks restarted at every peer segment
# (tuning.halo_bands) with the bf16 blocks allowed on a shard (dense3_min_blocks=0), against the shipped plan.  HIP-graph replay.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p5; rm -rf $out; mkdir -p $out
run() { n=$(echo "$1" | tr '/+ =,' '_-__.' | tr -s '_')_$2
  PGCN_TUNING="$1" timeout 400 python bench.py --emulate-rank 0/8 --graph --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python -c "
import json; r=json.load(open('$out/bench_$n.json')); g=r.get('graph_replay',{}); print('%-70s'%'[$1]', 'eager ms/epoch %.3f'%r['ms_per_step'], 'replay', g.get('ms_per_step'), 'halo groups', r.get('halo_groups'))" || tail -3 "$out/bench_$n.err"; }
for rep in 1 2; do
run "exchange_rounds=2" $rep
run "halo_bands=1,dense3_min_blocks=0" $rep
run "halo_bands=1,dense3_min_blocks=0,dense3_tau=0.12" $rep
run "halo_bands=1,dense3_min_blocks=0,dense3_tau=0.12,strip_min_records=0" $rep
run "halo_bands=0,dense3_min_blocks=0" $rep
done
