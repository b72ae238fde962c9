#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder_fwd_bwd or conv" 2>&1 | tail -3
for v in 65536 999999999; do echo "DEEPK_ROWS=$v: $(TACORL_CONV_DEEPK_ROWS=$v python scratch/run_configs.py c4real | tail -1)"; done
bash scratch/prof_cfg.sh c4real 13 2>&1 | sed -n 2,8p
