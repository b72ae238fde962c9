#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scratch/prof_cfg.sh c4real 13 2>&1 | tail -45
