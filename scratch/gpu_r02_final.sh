#!/bin/bash
# round-2 profiles: bench line, kernel stats, SQ counters and HBM traffic of the fused encoder inside bench.py
export TMPDIR=/tmp
O=gpurun_out/r02f; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-distribution > $O/bench_traced.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python scratch/stats_top.py $O/trace 23 > $O/stats_top.txt
rm -rf $O/trace
B="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-distribution"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc1 -- $B > /dev/null 2> $O/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc2 -- $B > /dev/null 2> $O/pmc2.err
python scratch/pmc_summary.py $O/pmc_sq.md $O/pmc1 $O/pmc2 --match "encoder_fused_kernel,ebw_" > /dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcf -- $B > /dev/null 2> $O/pmcf.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcw -- $B > /dev/null 2> $O/pmcw.err
python scratch/traffic_summary.py $O/fused_traffic.json $O/pmcf $O/pmcw "encoder_fused_kernel"
rm -rf $O/pmc1 $O/pmc2 $O/pmcf $O/pmcw
cat $O/bench_line.json | cut -c1-900; head -12 $O/stats_top.txt
