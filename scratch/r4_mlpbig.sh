#!/bin/bash
O=gpurun_out/r4mb; mkdir -p $O
python scratch/bench_mlp_big.py > $O/plain.txt 2>&1
TACORL_LIB=scratch/libs/stamps.so python scratch/bench_mlp_big.py > $O/stamps.txt 2>&1
cat $O/plain.txt $O/stamps.txt
