#!/bin/bash
# same-box A/B: working-tree fused encoder vs scratch/libs/ef_head.so (the previous commit's) + parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder" > gpurun_out/r6/enc_tests2.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r6/enc_tests2.log
SPEC="4096 512a 256 512a 512a 512 512"
for i in 1 2 3; do
  echo "tree: $(python scratch/run_fused.py $SPEC | head -1)"
  echo "head: $(TACORL_SCRATCH_LIB=scratch/libs/ef_head.so python scratch/run_fused.py $SPEC | head -1)"
done
for hw in 64; do
  echo "HW=$hw tree: $(HW=$hw python scratch/run_fused.py $SPEC | head -1)"
  echo "HW=$hw head: $(HW=$hw TACORL_SCRATCH_LIB=scratch/libs/ef_head.so python scratch/run_fused.py $SPEC | head -1)"
done
