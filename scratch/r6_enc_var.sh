#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SPEC="4096 512a 256 512a 512a 512 512"
echo "product: $(python scratch/run_fused.py $SPEC | head -1)"
for v in nodma v1 v2 v3 x2 x3 v3x1; do
echo "$v: $(TACORL_SCRATCH_LIB=scratch/libs/ef_$v.so python scratch/run_fused.py $SPEC | head -1)"
done
echo "product: $(python scratch/run_fused.py $SPEC | head -1)"
