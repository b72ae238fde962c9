#!/bin/bash
# SQ counters of the PlayLMP.training_step kernels (B=256, eager launches: two passes of 8 counters)
export TMPDIR=/tmp
O=gpurun_out/pmc_plmp; rm -rf $O; mkdir -p $O
C1="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"
C2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
ONLY=plmp NOGRAPH=1 STEPS=4 WARM=2 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C1 -d $O/p1 -- python3 scratch/run_c3.py > /dev/null 2> $O/p1.err
ONLY=plmp NOGRAPH=1 STEPS=4 WARM=2 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C2 -d $O/p2 -- python3 scratch/run_c3.py > /dev/null 2> $O/p2.err
python scratch/pmc_summary.py $O/pmc_sq_playlmp.md $O/p1 $O/p2 --match "rnn_gemm,rnn_wgrad,pr_encoder,pr_ln,wgrad_slab,logistic,ad_input" > /dev/null
rm -rf $O/p1 $O/p2
head -60 $O/pmc_sq_playlmp.md
