#!/bin/bash
# round-6 profiles in one GPU call: bench line, kernel stats + per-launch encoder durations (rocprofv3 of the bench command),
# SQ counters and HBM traffic of the fused encoder inside bench.py, the branch timeline from device time marks, kernel stats
# of C5 (many-row MLP kernels) and PlayLMP B=32 / B=256
export TMPDIR=/tmp
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-configs --no-cpu-baseline --condition-ms 0 > $O/bench_traced.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python scratch/encoder_launches.py $O/trace $O/encoder_launches.json > $O/encoder_launches.txt
python scratch/step_sequence.py $O/trace > $O/step_sequence.txt
rm -rf $O/trace
B="python3 bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-graph --no-distribution --condition-ms 0"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc1 -- $B > /dev/null 2> $O/pmc1.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc2 -- $B > /dev/null 2> $O/pmc2.err
python scratch/pmc_summary.py $O/pmc_sq.md $O/pmc1 $O/pmc2 --match "encoder_fused_kernel,ebw_,softargmax_bwd,mlp_,pr_encoder,rnn_gemm,prep_multi" > /dev/null
rm -rf $O/pmc1 $O/pmc2
T="python3 bench.py --steps 20 --warmup 3 --no-configs --no-cpu-baseline --no-distribution --no-graph --condition-ms 0"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/f -- $T > /dev/null 2> $O/f.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/w -- $T > /dev/null 2> $O/w.err
python scratch/traffic_summary.py $O/fused_traffic_raw.json $O/f $O/w encoder_fused_kernel > /dev/null
rm -rf $O/f $O/w
timeout 300 python scratch/marks2.py > $O/marks.txt 2>&1
# C5: kernel stats + SQ counters of the many-row MLP kernels
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5t -- python3 scratch/run_configs.py c5 > $O/c5_run.log 2> $O/c5.err
cp $(find $O/c5t -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv; rm -rf $O/c5t
NOGRAPH=1 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d $O/c5p1 -- python3 scratch/run_configs.py c5 > /dev/null 2> $O/c5p1.err
NOGRAPH=1 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/c5p2 -- python3 scratch/run_configs.py c5 > /dev/null 2> $O/c5p2.err
python scratch/pmc_summary.py $O/pmc_sq_c5.md $O/c5p1 $O/c5p2 --match "mlp_pers,mlp_big,mlp_mid,mlp_wgrad_big,mlp_wgrad_out,mlp_x_to" > /dev/null
rm -rf $O/c5p1 $O/c5p2
# C3 / C4 share / PlayLMP kernel stats
bash scratch/prof_c3.sh > $O/c3.txt 2>&1; cp gpurun_out/prof_c3/kernel_stats.csv $O/c3_kernel_stats.csv 2>/dev/null
bash scratch/prof_cfg.sh c4 13 > $O/c4.txt 2>&1; cp gpurun_out/prof_c4/kernel_stats.csv $O/c4_kernel_stats.csv 2>/dev/null
bash scratch/prof_cfg.sh c4real 13 > $O/c4real.txt 2>&1; cp gpurun_out/prof_c4real/kernel_stats.csv $O/c4real_kernel_stats.csv 2>/dev/null
bash scratch/prof_plmp.sh > $O/plmp.txt 2>&1; cp gpurun_out/prof_plmp/kernel_stats.csv $O/playlmp_kernel_stats.csv 2>/dev/null
cut -c1-400 $O/bench_line.json; cat $O/encoder_launches.txt; tail -3 $O/marks.txt; cat $O/fused_traffic_raw.json | head -12; tail -2 $O/c5_run.log
