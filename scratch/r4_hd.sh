#!/bin/bash
mkdir -p gpurun_out/r4hd
timeout 1500 python -m pytest tests -x -q -m gpu -k "tacorl_q_ad or tacorl_bc_ad or c3 or dist or rccl or step" > gpurun_out/r4hd/test.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4hd/test.txt
tail -4 gpurun_out/r4hd/test.txt
