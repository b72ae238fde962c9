#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x -k "playlmp" 2>&1 | tail -2
for B in 32 256; do timeout 300 python scratch/ab_plmp.py $B pp_gather False True 3 2>/dev/null | tail -2; done
