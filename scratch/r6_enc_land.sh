#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for S in "4096 512a 256 512a 512a 512 512" "4096 1536 1280"; do
for i in 1 2 3; do
echo "[$S] product:          $(python scratch/run_fused.py $S | head -1)"
echo "[$S] no landing wait:  $(TACORL_SCRATCH_LIB=scratch/libs/ef_p32.so python scratch/run_fused.py $S | head -1)"
done; done
