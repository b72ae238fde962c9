#!/bin/bash
# round-3 profiles in one GPU call: bench line, kernel stats + per-launch encoder durations (rocprofv3 of the bench
# command), SQ counters and HBM traffic of the fused encoder inside bench.py, the branch timeline from device time marks
export TMPDIR=/tmp
O=gpurun_out/r03f; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-configs --no-cpu-baseline > $O/bench_traced.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python scratch/encoder_launches.py $O/trace $O/encoder_launches.json > $O/encoder_launches.txt
python scratch/step_sequence.py $O/trace > $O/step_sequence.txt
rm -rf $O/trace
B="python3 bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-graph --no-distribution"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc1 -- $B > /dev/null 2> $O/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc2 -- $B > /dev/null 2> $O/pmc2.err
python scratch/pmc_summary.py $O/pmc_sq.md $O/pmc1 $O/pmc2 --match "encoder_fused_kernel,ebw_,softargmax_bwd" > /dev/null
rm -rf $O/pmc1 $O/pmc2
timeout 300 python scratch/marks2.py > $O/marks.txt 2>&1
cut -c1-600 $O/bench_line.json; cat $O/encoder_launches.txt; tail -3 $O/marks.txt; head -40 $O/pmc_sq.md
