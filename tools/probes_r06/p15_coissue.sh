# r06: VALU instructions under one bf16 MFMA, one and two waves per SIMD (tools/micro/mfma_valu_coissue.hip)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_p15; rm -rf $out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_coissue tools/micro/mfma_valu_coissue.hip && timeout 120 /tmp/mfma_valu_coissue | tee $out/coissue.txt
