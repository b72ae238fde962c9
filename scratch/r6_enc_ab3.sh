#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder" > gpurun_out/r6/enc_tests3.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r6/enc_tests3.log
SPEC="4096 512a 256 512a 512a 512 512"
for i in 1 2 3; do
  echo "tree (spread):  $(python scratch/run_fused.py $SPEC | head -1)"
  echo "nospread:       $(TACORL_SCRATCH_LIB=scratch/libs/ef_nospread.so python scratch/run_fused.py $SPEC | head -1)"
done
echo "head (deferral only): $(TACORL_SCRATCH_LIB=scratch/libs/ef_head.so python scratch/run_fused.py $SPEC | head -1)"
