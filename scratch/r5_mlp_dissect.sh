#!/bin/bash
# where the persistent many-row forward's time goes: full / no copy-out stores / no transcendental / neither
export TMPDIR=/tmp
O=gpurun_out/r5mlp; mkdir -p $O
for d in 0 1 2 3; do echo -n "dbg=$d  "; TACORL_MLP_PERS_DBG=$d timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1; done | tee $O/dissect.txt
echo -n "per-block kernels  "; TACORL_MLP_PERS=0 TACORL_MLP_PERS_BWD=0 timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1 | tee -a $O/dissect.txt
