#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SPEC="4096 512a 256 512a 512a 512 512"
TACORL_SCRATCH_LIB=scratch/libs/ef_blk.so python scratch/run_fused.py $SPEC | tail -3
for i in 1 2; do
echo "product (act 72, setup 128): $(python scratch/run_fused.py $SPEC | head -1)"
for v in act68 act76 set64 set192; do echo "$v: $(TACORL_SCRATCH_LIB=scratch/libs/ef_$v.so python scratch/run_fused.py $SPEC | head -1)"; done
done
