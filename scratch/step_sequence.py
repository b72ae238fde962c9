"""usage: step_sequence.py <dir with *kernel_trace.csv>: the kernel sequence of ONE training step (between two encoder
forward launches, late in the run) in start order with durations - rocprofv3 serialises graph branches, so this is the
work list of a step, not its timeline (scratch/marks.py gives that)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
enc = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void encoder_fused_kernel")]
# a step deep in the timed region: the 150th encoder launch
a, b = enc[150], enc[151]
t0 = int(rows[a]["Start_Timestamp"])
tot = 0.0
for r in rows[a - 6:b - 6]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:7.1f}  {n[:110]}")
print(f"kernel time of the step: {tot:.0f} us over {b - a} launches")
