#!/bin/bash
# round-2 session B: whole GPU suite (new parity tests), SQ counters for the encoder fwd + conv bwd kernels inside bench.py
export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
M="encoder_fused_kernel,ebw_"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-distribution > /dev/null 2> $O/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-distribution > /dev/null 2> $O/pmc2.err
python scratch/pmc_summary.py $O/pmc_sq.md $O/pmc1 $O/pmc2 --match $M > /dev/null
rm -rf $O/pmc1 $O/pmc2
tail -60 $O/tests.log
