"""Print the kernel timeline of one replayed step from a rocprofv3 --kernel-trace csv.
usage: python scratch/timeline.py <dir with *_kernel_trace.csv> [step_index_from_end]"""
import csv, glob, sys, re
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n)
    return n[:70]
# a step starts with the image pack kernel
marks = [i for i, r in enumerate(rows) if "pack_nchw3_batch" in r["Kernel_Name"]]
s, e = marks[-back - 1], marks[-back]
t0 = int(rows[s]["Start_Timestamp"]); last_end = t0
print(f"step span {(int(rows[e]['Start_Timestamp']) - t0) / 1e3:.1f} us, {e - s} kernels")
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(st - t0) / 1e3:8.1f} {(en - st) / 1e3:7.1f} q{r.get('Queue_Id', '?'):>2} s{r.get('Stream_Id', '?'):>3} "
          f"g{r.get('Grid_Size', '?'):>8} {short(r['Kernel_Name'])}")
