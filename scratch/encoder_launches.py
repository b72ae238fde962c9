"""usage: encoder_launches.py <dir with *kernel_trace.csv> <out.json>: durations of the encoder-forward launches of a
rocprofv3 --kernel-trace run of bench.py, in time order - inside the steps vs the back-to-back probe at the end."""
import csv
import glob
import json
import statistics as st
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void encoder_fused_kernel")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
t0 = int(rows[0]["Start_Timestamp"])
gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
# back-to-back probe launches follow their predecessor within a few us; in-step launches are a step apart
b2b = [d[i + 1] for i, g in enumerate(gap) if g < 30]
step = [d[i + 1] for i, g in enumerate(gap) if g >= 300]
print(f"encoder_fused_kernel launches: {len(d)}; in steps {len(step)}: mean {st.mean(step):.1f} median {st.median(step):.1f} us; "
      f"back to back {len(b2b)}: mean {st.mean(b2b):.1f} median {st.median(b2b):.1f} us; all: mean {st.mean(d):.1f} us")
json.dump({"in_step_us": {"n": len(step), "mean": st.mean(step), "median": st.median(step)},
           "back_to_back_us": {"n": len(b2b), "mean": st.mean(b2b) if b2b else None},
           "all_mean_us": st.mean(d)}, open(sys.argv[2], "w"), indent=1)
