#!/bin/bash
# round 4: C5 many-row MLP path - kernel tests, full-size C5 tests, timing with and without the LDS-DMA weight gradients, kernel stats
export TMPDIR=/tmp
O=gpurun_out/r4c5; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "many_rows or lean or mlp_fused" > $O/kern.txt 2>&1; echo "kern rc=$?" >> $O/kern.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "c5" > $O/full.txt 2>&1; echo "full rc=$?" >> $O/full.txt
TACORL_MLP_BIG_WGRAD=0 timeout 300 python scratch/run_configs.py c5 > $O/c5_old.txt 2>&1
timeout 300 python scratch/run_configs.py c5 > $O/c5_new.txt 2>&1
bash scratch/prof_cfg.sh c5 25 > $O/prof.txt 2>&1
cp gpurun_out/prof_c5/kernel_stats.csv $O/kernel_stats.csv
tail -4 $O/kern.txt $O/full.txt; cat $O/c5_old.txt $O/c5_new.txt | grep C5; head -24 $O/prof.txt
