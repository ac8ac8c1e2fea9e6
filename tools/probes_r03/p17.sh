#!/bin/bash
# r03 batch 17: HIP-graph replay of one training step (N = 1 and emulated ranks)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r03_p17; rm -rf $out; mkdir -p $out
for a in "" "--emulate-rank 0/8"; do
  t=$(echo "n1$a" | tr -d ' /-')
  python bench.py $a --graph --steps 10 --warmup 3 --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python -c "
import json; r=json.load(open('$out/b_$t.json')); print('$a', 'eager %.3f ms' % r['ms_per_step'], r.get('graph_replay'))" || tail -5 $out/b_$t.err
done
