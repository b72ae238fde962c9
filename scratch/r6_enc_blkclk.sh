#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SPEC="4096 512a 256 512a 512a 512 512"
for v in ef_blk ef_blk_l2 ef_blk_nodma; do
  echo "== $v"; TACORL_SCRATCH_LIB=scratch/libs/$v.so python scratch/run_fused.py $SPEC | grep -v "^$"
done
