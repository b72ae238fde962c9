#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for cus in 256 248 240 224; do
  for i in 1 2; do echo "PERS_CUS=$cus: $(TACORL_MLP_PERS_CUS=$cus python scratch/run_configs.py c5 | tail -1)"; done
done
