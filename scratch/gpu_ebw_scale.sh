#!/bin/bash
# per-kernel durations of the fused encoder backward at 1, 2, 6, 12 images per workgroup (fixed cost vs per-image cost)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 85 170 512 1020; do
  rm -rf /tmp/ebw_$n
  NIMG=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ebw_$n -o p -- python3 $R/scratch/run_ebw.py 20 > /dev/null 2>&1
  f=$(find /tmp/ebw_$n -name "*kernel_stats.csv" | head -1)
  echo "== NIMG=$n" >> $R/gpurun_out/ebw_scale.txt
  python3 - "$f" >> $R/gpurun_out/ebw_scale.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Name"]
    if any(k in nm for k in ("ebw_", "softargmax_bwd", "sum_to_scalar", "mlp_", "copy_cols", "rows")):
        print(f'{float(r["AverageNs"])/1e3:8.1f} us  x{r["Calls"]:>4}  {nm[:110]}')
PY
done
