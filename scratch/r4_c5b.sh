#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c5; mkdir -p $O
python scratch/bench_mlp_big.py > $O/mb_plain.txt 2>&1
TACORL_LIB=scratch/libs/stamps.so python scratch/bench_mlp_big.py > $O/mb_stamps.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "mlp" > $O/kern.txt 2>&1; echo "kern rc=$?" >> $O/kern.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "c5" > $O/full.txt 2>&1; echo "full rc=$?" >> $O/full.txt
TACORL_MLP_BIG=0 timeout 300 python scratch/run_configs.py c5 > $O/c5_old.txt 2>&1
timeout 300 python scratch/run_configs.py c5 > $O/c5_new.txt 2>&1
cat $O/mb_plain.txt $O/mb_stamps.txt; tail -n 4 $O/kern.txt $O/full.txt; cat $O/c5_old.txt $O/c5_new.txt | grep C5
