# per-step kernel list of the default (GCN) bench: what an epoch launches beside the six aggregation groups
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p16; rm -rf $out; mkdir -p $out
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o gcn -- python $GRAFT_REPO_ROOT/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$out/prof_stdout.log 2> $GRAFT_REPO_ROOT/$out/prof_stderr.log
cd $GRAFT_REPO_ROOT; rm -f $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv; ls $out/prof
