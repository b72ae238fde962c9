"""usage: stats_top.py <dir with *kernel_stats.csv> <steps>  -> per-step kernel time of the top kernels"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows) / steps / 1e3
print(f"total kernel time per step: {tot:.0f} us")
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    print(f"{int(r['Calls']) / steps:6.1f}/step {int(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step  avg {float(r['AverageNs']) / 1e3:7.1f}  {n[:100]}")
