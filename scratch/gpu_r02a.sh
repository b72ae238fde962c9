#!/bin/bash
# round-2 session A: new tests, bench line, kernel stats, SQ counters for the encoder fwd + conv bwd kernels inside bench.py
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "hipgraph or ranks" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-distribution > $O/bench_traced.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python scratch/stats_top.py $O/trace 23 > $O/stats_top.txt
rm -rf $O/trace
M="encoder_fused_kernel,ebw_"
timeout 600 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-distribution > /dev/null 2> $O/pmc1.err
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-distribution > /dev/null 2> $O/pmc2.err
python scratch/pmc_summary.py $O/pmc_sq.md $O/pmc1 $O/pmc2 --match $M > /dev/null
rm -rf $O/pmc1 $O/pmc2
tail -5 $O/tests.log; cat $O/bench.json; cat $O/stats_top.txt
