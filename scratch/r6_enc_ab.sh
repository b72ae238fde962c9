#!/bin/bash
# same-box A/B of the fused encoder forward: product library (leftover-pixel deferral) vs -DEF_DEFER=0
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SPEC="4096 512a 256 512a 512a 512 512"
for i in 1 2 3; do
  echo "defer:   $(python scratch/run_fused.py $SPEC | head -1)"
  echo "nodefer: $(TACORL_SCRATCH_LIB=scratch/libs/ef_nodefer.so python scratch/run_fused.py $SPEC | head -1)"
done
echo "C5-like (11264 images, 6144 with activations)"
for i in 1 2; do
  echo "defer:   $(python scratch/run_fused.py 2048a 1024 2048a 2048a 2048 2048 | head -1)"
  echo "nodefer: $(TACORL_SCRATCH_LIB=scratch/libs/ef_nodefer.so python scratch/run_fused.py 2048a 1024 2048a 2048a 2048 2048 | head -1)"
done
