"""usage: traffic_summary.py <out.json> <fetch_dir> <write_dir> <kernel substring> : mean FETCH_SIZE / WRITE_SIZE (KiB) per
dispatch of one kernel from two separate rocprofv3 --pmc passes; gfx950 correction (FETCH_SIZE x2) applied."""
import csv, glob, json, sys
out, fd, wd, name = sys.argv[1:5]
csv.field_size_limit(1 << 30)
def mean(d, counter):
    per = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if name in k and "pr_" + name not in k and r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = list(per.values())
    return sum(v) / len(v), len(v)
f, nf = mean(fd, "FETCH_SIZE")
w, nw = mean(wd, "WRITE_SIZE")
res = {"kernel": name, "dispatches": [nf, nw], "FETCH_SIZE_KiB_raw": round(f), "WRITE_SIZE_KiB": round(w),
       "hbm_read_bytes_corrected": int(f * 1024 * 2), "hbm_write_bytes": int(w * 1024),
       "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024)}
json.dump(res, open(out, "w"), indent=1)
print(res)
