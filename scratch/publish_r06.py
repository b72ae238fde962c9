"""Round-6 profiles: gpurun_out/r06f (scratch/gpu_r06_final.sh) + gpurun_out/r5suite (the GPU suite's parity margins) ->
profiles/r06_*.  usage: python scratch/publish_r06.py [tag]   (tag: suffix of the published files, default none)"""
import hashlib, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, P = os.path.join(ROOT, "gpurun_out", "r06f"), os.path.join(ROOT, "profiles")
cp = lambda a, b: shutil.copy(os.path.join(S, a), os.path.join(P, b))  # noqa: E731
cp("bench_line.json", "r06_bench_line.json")
cp("bench_traced.json", "r06_bench_line_under_rocprof.json")
cp("kernel_stats.csv", "r06_bench_kernel_stats.csv")
cp("encoder_launches.json", "r06_encoder_launches.json")
cp("step_sequence.txt", "r06_step_sequence.txt")
cp("pmc_sq.md", "r06_pmc_sq.md")
cp("pmc_sq_c5.md", "r06_pmc_sq_c5.md")
for k in ("c3", "c4", "c4real", "c5", "playlmp"):
    if os.path.exists(os.path.join(S, f"{k}_kernel_stats.csv")):
        cp(f"{k}_kernel_stats.csv", f"r06_{k}_kernel_stats.csv")
line = json.load(open(os.path.join(S, "bench_line.json")))
raw = json.load(open(os.path.join(S, "fused_traffic_raw.json")))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
sha = bench.kernel_source_hash()  # (encoder_fused.hip + encoder_fused.h)
traffic = {"encoder_fused_hip_sha256": sha,
           "bench_launch": dict(raw, images_per_launch=6912,
                                algorithmic_bytes={"images_bf16": 6912 * 42336, "outputs_f32": 6912 * 128, "saved_activations_1536_images": 77800000},
                                how="rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a second pass, --pmc WRITE_SIZE of `python3 bench.py --steps 20 "
                                    "--warmup 3 --no-configs --no-cpu-baseline --no-distribution --no-graph` (scratch/gpu_r06_final.sh), mean over the "
                                    "encoder_fused_kernel<84,84> dispatches; FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, "
                                    "MI355X_MICROARCH.md HBM section).  bench.py reports `traffic` only while encoder_fused_hip_sha256 matches the tree.")}
json.dump(traffic, open(os.path.join(P, "r06_fused_traffic.json"), "w"), indent=1)
marks = open(os.path.join(S, "marks.txt")).read().strip().splitlines()
tl = dict(kv.split("=") for kv in marks[-1].split())
untraced = [l for l in marks if "traced=False" in l][-1].split(":")[-1].strip()
traced = [l for l in marks if "traced=True" in l][-1].split(":")[-1].strip()
r4 = {}
try:
    for row in open(os.path.join(P, "r05_step_timeline.md")):
        c = [x.strip() for x in row.split("|")]
        if len(c) >= 3 and c[2].isdigit():
            r4[c[1]] = int(c[2])
except OSError:
    pass
with open(os.path.join(P, "r06_step_timeline.md"), "w") as f:
    f.write("# Branch timeline of the headline step, round 6 (device time marks, `scratch/marks2.py`)\n\n"
            "1-thread launches that store the device clock sit between the captured launches of the step; the graph is replayed as usual\n"
            f"(the marks add launches: {traced} traced against {untraced} untraced on the same box - read the intervals).  Microseconds since\n"
            "the first mark of the graph (the eager image pack, ~105 us, runs in front of it); round 5 in brackets (`r05_step_timeline.md`).\n\n"
            "| mark | us | (round 5) |\n|---|---|---|\n")
    for k, v in tl.items():
        f.write(f"| {k} | {v} | {r4.get(k, '')} |\n")
    f.write(f"\nMain chain `b:critic_bwd -> b:enc_bwd` {int(tl['b:enc_bwd']) - int(tl['b:critic_bwd'])} us; action-decoder branch "
            f"`ad:start -> ad:end` {int(tl['ad:end']) - int(tl['ad:start'])} us; `c:adam` {tl['c:adam']} vs `ad:end` {tl['ad:end']}.\n")
mj = os.path.join(ROOT, "gpurun_out", "parity_margins.jsonl")
if os.path.exists(mj):
    subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "margins_md.py"), mj],
                   stdout=open(os.path.join(P, "r06_parity_margins.md"), "w"), check=True)
c = line["configs"]
json.dump({"c2_three_segment": c.get("c2_three_segment"), "c3_strong_share_b32": c.get("c3_strong_share_b32"),
           "command": "python bench.py (default run; the probe is the child process `python bench.py --probe segments`)",
           "headline_ms_per_step_same_run": line["ms_per_step"]}, open(os.path.join(P, "r06_segment_probe.json"), "w"), indent=1)
print("published; headline", line["ms_per_step"], "ms/step, roofline", line["roofline"]["frac"])
