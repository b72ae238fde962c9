#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab14; mkdir -p $O
TACORL_EF_SPLIT_LMP=1 timeout 900 python -m pytest tests/test_step_gpu.py -q -m gpu -x -k "tacorl_step" 2>&1 | tail -2
for cfg in "0 192" "1 192" "1 224" "1 256" "0 192" "1 192"; do set -- $cfg; echo -n "SPLIT=$1 BUDGET=$2  "; TACORL_EF_SPLIT_LMP=$1 TACORL_EF_SPLIT_BUDGET=$2 timeout 300 python scratch/run_configs.py c4 2>/dev/null | tail -1; done | tee $O/c4_split.txt
timeout 300 python scratch/ab_step.py env:TACORL_EF_SPLIT_LMP 0 1 3 2>/dev/null | tail -2 | tee -a $O/c4_split.txt
TACORL_EF_SPLIT_BUDGET=224 timeout 300 python scratch/ab_step.py env:TACORL_EF_SPLIT_LMP 0 1 2 2>/dev/null | tail -2 | tee -a $O/c4_split.txt
TACORL_EF_SPLIT_LMP=1 MARKS=1 timeout 300 python scratch/run_configs.py c4 2>/dev/null | tail -2 | head -1
