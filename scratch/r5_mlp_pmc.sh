#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5mlp; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/p1 -- python3 scratch/bench_mlp_big.py > /dev/null 2> $O/p1.err
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/p2 -- python3 scratch/bench_mlp_big.py > /dev/null 2> $O/p2.err
python scratch/pmc_summary.py $O/pmc_mlp_pers.md $O/p1 $O/p2 --match "mlp_pers,mlp_wgrad_big,mlp_wgrad_out,mlp_x_to" > /dev/null
rm -rf $O/p1 $O/p2; cat $O/pmc_mlp_pers.md | cut -c1-250
