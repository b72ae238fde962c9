#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab9; mkdir -p $O
timeout 1200 python -m pytest tests/test_step_gpu.py -q -m gpu -x -k "tacorl" 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python scratch/ab_step.py env:TACORL_AD_LOSS_LAZY 0 1 3 2>/dev/null | tail -2 | tee $O/ab_lazy.txt
