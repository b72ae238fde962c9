#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SPEC="4096 512a 256 512a 512a 512 512"
echo "== stamps, deferral"; TACORL_SCRATCH_LIB=scratch/libs/ef_stamps.so python scratch/run_fused.py $SPEC
echo "== stamps, no deferral"; TACORL_SCRATCH_LIB=scratch/libs/ef_stamps_nd.so python scratch/run_fused.py $SPEC
echo "== no in-loop DMA (stale images), deferral"; TACORL_SCRATCH_LIB=scratch/libs/ef_nodma.so python scratch/run_fused.py $SPEC | head -1
echo "== product"; python scratch/run_fused.py $SPEC | head -1
