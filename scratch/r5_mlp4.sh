#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5mlp; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "mlp_persistent" 2>&1 | tail -2
for v in 1 1 1; do echo -n "PERS=$v  "; TACORL_MLP_PERS=$v timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1; done | tee $O/packed.txt
