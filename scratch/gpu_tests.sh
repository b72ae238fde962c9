#!/bin/bash
# usage: gpu_tests.sh <tag> <pytest args...>   full log -> gpurun_out/tests_<tag>.log, digest on stdout
tag=$1; shift
mkdir -p gpurun_out
timeout 2400 python -m pytest "$@" > gpurun_out/tests_$tag.log 2>&1
echo "rc=$?" >> gpurun_out/tests_$tag.log
grep -E "^(E   |FAILED|ERROR|[0-9]+ (passed|failed)|rc=)" gpurun_out/tests_$tag.log | cut -c1-240 | head -150
