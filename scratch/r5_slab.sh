#!/bin/bash
mkdir -p gpurun_out/r5slab; O=gpurun_out/r5slab
for B in 256 32; do for r in 1 2; do for w in 256 1024 4096; do
echo "wgs=$w $(TACORL_SLAB_REDUCE_WGS=$w timeout 200 python scratch/ab_plmp.py $B dummy 0 1 1 2>&1 | tail -2 | tr '\n' ' ')"
done; done; done | tee $O/caps.txt
