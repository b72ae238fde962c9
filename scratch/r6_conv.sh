#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder_fwd_bwd or conv or encoder_fused_forward" > gpurun_out/r6/conv_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r6/conv_tests.log
python scratch/run_configs.py c4real | tail -1
bash scratch/prof_cfg.sh c4real 13 2>&1 | sed -n 2,14p
