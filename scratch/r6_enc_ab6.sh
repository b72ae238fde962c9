#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder" > gpurun_out/r6/enc_tests6.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r6/enc_tests6.log
SPEC="4096 512a 256 512a 512a 512 512"
for i in 1 2 3; do
  echo "tree: $(python scratch/run_fused.py $SPEC | head -1)"
  echo "head: $(TACORL_SCRATCH_LIB=scratch/libs/ef_head.so python scratch/run_fused.py $SPEC | head -1)"
done
echo "== blk clocks"; TACORL_SCRATCH_LIB=scratch/libs/ef_blk.so python scratch/run_fused.py $SPEC
echo "C5-like tree: $(python scratch/run_fused.py 2048a 1024 2048a 2048a 2048 2048 | head -1)"
echo "C5-like head: $(TACORL_SCRATCH_LIB=scratch/libs/ef_head.so python scratch/run_fused.py 2048a 1024 2048a 2048a 2048 2048 | head -1)"
echo "128 tree: $(HW=128 python scratch/run_fused.py 2048 128a 64 128a 128a 128 128 2048 128a 64 128a 128a 128 128| head -1)"
echo "128 head: $(HW=128 TACORL_SCRATCH_LIB=scratch/libs/ef_head.so python scratch/run_fused.py 2048 128a 64 128a 128a 128 128 2048 128a 64 128a 128a 128 128 | head -1)"
