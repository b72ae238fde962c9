#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5mlp; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "mlp_persistent or many_rows" 2>&1 | tail -2
for d in 0 1 0; do echo -n "dbg=$d  "; TACORL_MLP_PERS_DBG=$d timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1; done | tee $O/dissect2.txt
