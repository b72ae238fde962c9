#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SPEC="4096 512a 256 512a 512a 512 512"
for i in 1 2; do
  echo "product:          $(python scratch/run_fused.py $SPEC | head -1)"
  echo "same image (L2):  $(TACORL_SCRATCH_LIB=scratch/libs/ef_l2.so python scratch/run_fused.py $SPEC | head -1)"
  echo "no DMA:           $(TACORL_SCRATCH_LIB=scratch/libs/ef_nodma.so python scratch/run_fused.py $SPEC | head -1)"
done
