#!/bin/bash
export TACORL_DIST_BACKEND=gloo TACORL_BENCH_SINGLE_DEVICE=1 MASTER_ADDR=127.0.0.1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --batch 64 --no-cpu-baseline > gpurun_out/dist_dbg.out 2> gpurun_out/dist_dbg.err
echo rc=$? >> gpurun_out/dist_dbg.out
