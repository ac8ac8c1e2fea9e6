# Cache policy of the streamed-once operands (strip records, bf16-block A values) and of the strip panels: `nt` variants of the library
# (tools/ab_build.sh nt1 / nt2 / nt3) against the default build, Reddit shape twice each + the SBM stand-in once.
# gpurun --timeout 700 -- 'bash tools/probes_r05/p4_nt_policy.sh'
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
PKG=scalable-graph-convolutional-network-training-on-distributed-memory-systems_amd
out=gpurun_out/r05_p4; rm -rf $out; mkdir -p $out
cp $PKG/lib/libpgcn_hip.so $PKG/lib/libpgcn_hip.base.so
run() { # lib tag, label, extra bench args
  cp $PKG/lib/libpgcn_hip.$1.so $PKG/lib/libpgcn_hip.so
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline $3 > "$out/bench_$1_$2.json" 2> "$out/bench_$1_$2.err"
  python -c "
import json; r=json.load(open('$out/bench_$1_$2.json')); ro=r['roofline']; print('%-6s %-10s'%('$1','$2'), 'ms/epoch %.3f'%r['ms_per_step'], 'group fwd %.4f bwd %.4f'%(ro['avg_launch_ms'], ro.get('avg_launch_ms_backward_AT',0)), {k:int(v) for k,v in ro.get('split_us',{}).items() if isinstance(v,(int,float))})" || tail -3 "$out/bench_$1_$2.err"; }
for rep in 1 2; do for v in base nt1 nt2 nt3; do run $v reddit$rep ""; done; done
for v in base nt1 nt2 nt3; do run $v sbm "--generator sbm"; done
cp $PKG/lib/libpgcn_hip.base.so $PKG/lib/libpgcn_hip.so
