#!/bin/bash
# where the persistent many-row forward's time goes (a -DMLP_PERS_DBG scratch build: scratch/mklib_file.sh pers_dbg mlp_fused.hip -DMLP_PERS_DBG):
# TACORL_MLP_PERS_DBG bits: 1 = no copy-out stores, 2 = no transcendental, 4 = no MFMA loops, 8 = no epilogue
export TMPDIR=/tmp
O=gpurun_out/r5mlp; mkdir -p $O
for d in 0 1 2 5 9 13; do echo -n "dbg=$d  "; TACORL_HIP_LIB=scratch/libs/pers_dbg.so TACORL_MLP_PERS_DBG=$d timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1; done | tee $O/dissect.txt
echo -n "per-block kernels  "; TACORL_MLP_PERS=0 TACORL_MLP_PERS_BWD=0 timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1 | tee -a $O/dissect.txt
