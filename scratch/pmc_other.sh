#!/bin/bash
# SQ counters of the round-2 kernels outside the headline step: rnn_wgrad / BPTT wavefront (PlayLMP step) and the banded
# 128x128 encoder forward / backward (C4-like step).  Eager launches (no graph) so every dispatch is counted.
export TMPDIR=/tmp
O=gpurun_out/pmc_other; rm -rf $O; mkdir -p $O
C1="SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"
C2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
NOGRAPH=1 ONLY=plmp timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C1 -d $O/a1 -- python3 scratch/run_c3.py > /dev/null 2> $O/a1.err
NOGRAPH=1 ONLY=plmp timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C2 -d $O/a2 -- python3 scratch/run_c3.py > /dev/null 2> $O/a2.err
python scratch/pmc_summary.py $O/pmc_playlmp.md $O/a1 $O/a2 --match "rnn_wgrad_kernel,rnn_gemm_kernel,gemm_kernel" > /dev/null
NOGRAPH=1 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C1 -d $O/b1 -- python3 scratch/run_configs.py c4 > /dev/null 2> $O/b1.err
NOGRAPH=1 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C2 -d $O/b2 -- python3 scratch/run_configs.py c4 > /dev/null 2> $O/b2.err
python scratch/pmc_summary.py $O/pmc_c4.md $O/b1 $O/b2 --match "encoder_fused_kernel,ebw_,softargmax" > /dev/null
rm -rf $O/a1 $O/a2 $O/b1 $O/b2
head -30 $O/pmc_playlmp.md; head -30 $O/pmc_c4.md
