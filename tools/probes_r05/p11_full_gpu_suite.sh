# the whole GPU suite + smoke on a fresh box (mid-round check after the r05 changes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r05_p11; rm -rf $out; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q -x > $out/pytest_gpu_full.txt 2>&1; tail -8 $out/pytest_gpu_full.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.txt
