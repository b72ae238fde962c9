#!/bin/bash
# A/B of the conv-backward variants at full size (3 x 512 images): per-kernel durations under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ebw_ab.txt; rm -f $OUT
for v in "$@"; do
  rm -rf /tmp/ebw_$v
  env_fuse3=${v%%_*}
  TACORL_EBW_FUSE3=$env_fuse3 NIMG=512 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ebw_$v -o p -- python3 $R/scratch/run_ebw.py 20 > /tmp/ebw_$v.out 2>&1
  f=$(find /tmp/ebw_$v -name "*kernel_stats.csv" | head -1)
  echo "== FUSE3=$v  $(tail -1 /tmp/ebw_$v.out)" >> $OUT
  python3 - "$f" >> $OUT <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Name"]
    if any(k in nm for k in ("ebw_", "softargmax_bwd", "sum_to_scalar")) and "pack" not in nm:
        print(f'{float(r["AverageNs"])/1e3:8.1f} us  x{r["Calls"]:>4}  {nm[:110]}'); tot += float(r["AverageNs"])/1e3
print(f'{tot:8.1f} us  total conv chain')
PY
done
