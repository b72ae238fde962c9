#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab5; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "mlp_persistent or many_rows or gathered_input or mlp_fused" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
TACORL_MLP_PERS_BWD=0 timeout 300 python scratch/run_configs.py c5 > $O/c5_old.txt 2>&1; tail -1 $O/c5_old.txt
TACORL_MLP_PERS_BWD=1 timeout 300 python scratch/run_configs.py c5 > $O/c5_new.txt 2>&1; tail -1 $O/c5_new.txt
TACORL_MLP_PERS_BWD=0 timeout 300 python scratch/run_configs.py c5 > $O/c5_old2.txt 2>&1; tail -1 $O/c5_old2.txt
TACORL_MLP_PERS_BWD=1 timeout 300 python scratch/run_configs.py c5 > $O/c5_new2.txt 2>&1; tail -1 $O/c5_new2.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "c5" > $O/tests2.txt 2>&1; tail -3 $O/tests2.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5t -- python3 scratch/run_configs.py c5 > $O/c5_run.log 2> $O/c5.err
cp $(find $O/c5t -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv; rm -rf $O/c5t; head -8 $O/c5_kernel_stats.csv | cut -c1-150
