#!/bin/bash
export TMPDIR=/tmp
for cfg in "0 0" "1 1" "1 1" "1 0" "1 1" "0 0" "1 1"; do set -- $cfg; echo -n "SPLIT=$1 PRIO=$2  "; TACORL_EF_SPLIT_LMP=$1 TACORL_PR_PRIO=$2 timeout 300 python scratch/run_configs.py c4 2>/dev/null | tail -1; done
