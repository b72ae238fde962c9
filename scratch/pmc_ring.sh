#!/bin/bash
# SQ counters of encoder_ring_kernel<150,200> (two passes of 8) on the step's launch shape of experiment=tacorl_real_world at B = 64:
# 2 048 LMP-window images + 5 no-grad problems + 3 problems that save activations -> gpurun_out/pmc_ring/pmc_sq_ring.md
export TMPDIR=/tmp
O=gpurun_out/pmc_ring; rm -rf $O; mkdir -p $O
export HxW=150x200
A="scratch/run_fused.py 2048 64 64 64 64 64 128a 128a 128a"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/p1 -- python3 $A > /dev/null 2> $O/p1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/p2 -- python3 $A > /dev/null 2> $O/p2.err
python scratch/pmc_summary.py $O/pmc_sq_ring.md $O/p1 $O/p2 --match "encoder_ring_kernel" > /dev/null
rm -rf $O/p1 $O/p2
cat $O/pmc_sq_ring.md | head -40
