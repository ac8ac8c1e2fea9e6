#!/bin/bash
# r02 probe 4: does the 4 KB stride of col % 8 slicing cost L2 capacity?  range slices (contiguous rows per XCD)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r02_p4; rm -rf $out; mkdir -p $out
for K in 16384 32768 65536 131072; do
  echo "== uniform:$K"
  timeout 300 python tools/spmm_probe.py --workload uniform:$K --rounds 5 --variants s8c1024,r:s8c1024 2>&1 | grep -v amdgpu.ids | tee -a $out/uniform.txt
done
echo "== reddit"
timeout 600 python tools/spmm_probe.py --rounds 6 --split --variants s8c1024k,d:s8c1024k,d:s8c1024k0.02 2>&1 | grep -v amdgpu.ids | tee $out/reddit.txt
