#!/bin/bash
# the fused conv3 kernel with parts switched off (TACORL_L3_DBG) and at 1 / 6 / 12 images per workgroup
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/l3_dbg.txt; rm -f $OUT
for cfg in "0 512" "0 85" "0 1020" "7 512"; do
  set -- $cfg
  rm -rf /tmp/l3p
  TACORL_L3_DBG=$1 NIMG=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/l3p -o p -- python3 $R/scratch/run_ebw.py 20 > /tmp/l3p.out 2>&1
  f=$(find /tmp/l3p -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$1" "$2" >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ebw_l3" in r["Name"]:
        print(f'dbg={sys.argv[2]} nimg={sys.argv[3]}: {float(r["AverageNs"])/1e3:8.1f} us')
PY
done
