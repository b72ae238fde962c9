#!/bin/bash
# the GPU suite three times WITHOUT the release-after-every-test fixture (TACORL_TEST_NO_RELEASE=1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
for i in 1 2 3; do
  TACORL_TEST_NO_RELEASE=1 timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r6/norelease_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 gpurun_out/r6/norelease_$i.log)"
done
