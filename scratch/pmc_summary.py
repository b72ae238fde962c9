"""usage: pmc_summary.py <out.md> <dir> [<dir> ...] [--match substr,substr]
Per-kernel mean-per-dispatch table of every counter found in rocprofv3 --pmc output directories
(*counter_collection.csv).  Runs on the GPU box right after the passes so only the summary travels back."""
import collections
import csv
import glob
import re
import sys

args = sys.argv[1:]
match = None
if "--match" in args:
    i = args.index("--match")
    match = args[i + 1].split(",")
    del args[i:i + 2]
out, dirs = args[0], args[1:]


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
csv.field_size_limit(1 << 30)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)  # (dispatch, counter) -> value summed over rows (XCC instances)
        name_of = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if match and not any(m in k for m in match):
                continue
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            name_of[r["Dispatch_Id"]] = k
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"])
        for (disp, cn), v in per.items():
            acc[name_of[disp]][cn].append(v)
with open(out, "w") as w:
    for k in sorted(acc):
        g = meta[k]
        w.write(f"### {k}\n\ngrid {g[0]}, workgroup {g[1]}, LDS {g[2]} B, VGPR {g[3]} + AGPR {g[4]}, scratch {g[5]}\n\n| counter | mean per dispatch | dispatches |\n|---|---|---|\n")
        for cn in sorted(acc[k]):
            v = acc[k][cn]
            w.write(f"| {cn} | {sum(v) / len(v):.4g} | {len(v)} |\n")
        w.write("\n")
print(open(out).read()[:6000])
