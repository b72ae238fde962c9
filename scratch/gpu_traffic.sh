#!/bin/bash
# HBM traffic of the encoder-forward launch of the bench step: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
# passes of the same command (MI355X_MICROARCH.md: the two do not fit one pass), gfx950 correction (FETCH_SIZE x 2).
export TMPDIR=/tmp
O=gpurun_out/traffic; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/f -- python3 bench.py --steps 20 --warmup 3 --no-configs --no-cpu-baseline --no-distribution --no-graph > /dev/null 2> $O/f.err
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/w -- python3 bench.py --steps 20 --warmup 3 --no-configs --no-cpu-baseline --no-distribution --no-graph > /dev/null 2> $O/w.err
python scratch/traffic_summary.py $O/fused_traffic.json $O/f $O/w encoder_fused_kernel
rm -rf $O/f $O/w
