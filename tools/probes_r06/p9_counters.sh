# r06: does this image's rocprofv3 expose a memory-side counter (HBM / MALL) beside FETCH_SIZE / WRITE_SIZE?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p9; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$out/counters.txt 2>&1)
grep -c "" $out/counters.txt
grep -i -o "name:[ ]*[A-Za-z0-9_]*" $out/counters.txt | sort -u | grep -i "mall\|hbm\|dram\|umc\|EA0\|_EA_\|RDREQ\|WRREQ\|IO_\|GMI\|FETCH\|WRITE_SIZE\|MEM_\|MC_" | head -80
timeout 900 python -m pytest tests/test_gat_gpu.py -m gpu -q -x -k "rank_of_four" 2>&1 | tail -3
