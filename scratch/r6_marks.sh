#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python scratch/marks2.py 2>&1 | tail -4
