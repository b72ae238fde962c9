"""Round-4 profiles: gpurun_out/r04f (scratch/gpu_r04_final.sh) + gpurun_out/r4suite (the GPU suite's parity margins) ->
profiles/r04_*, and the {PLACEHOLDER}s of DESIGN.md.  usage: python scratch/publish_r04.py"""
import hashlib, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, P = os.path.join(ROOT, "gpurun_out", "r04f"), os.path.join(ROOT, "profiles")
cp = lambda a, b: shutil.copy(os.path.join(S, a), os.path.join(P, b))  # noqa: E731
cp("bench_line.json", "r04_bench_line.json")
cp("bench_traced.json", "r04_bench_line_under_rocprof.json")
cp("kernel_stats.csv", "r04_bench_kernel_stats.csv")
cp("encoder_launches.json", "r04_encoder_launches.json")
cp("step_sequence.txt", "r04_step_sequence.txt")
cp("pmc_sq.md", "r04_pmc_sq.md")
cp("pmc_sq_c5.md", "r04_pmc_sq_c5.md")
for k in ("c3", "c4", "c5", "playlmp"):
    cp(f"{k}_kernel_stats.csv", f"r04_{k}_kernel_stats.csv")
line = json.load(open(os.path.join(S, "bench_line.json")))
raw = json.load(open(os.path.join(S, "fused_traffic_raw.json")))
sha = hashlib.sha256(open(os.path.join(ROOT, "tacorl_amd", "csrc", "encoder_fused.hip"), "rb").read()).hexdigest()
traffic = {"encoder_fused_hip_sha256": sha,
           "bench_launch": dict(raw, images_per_launch=6912,
                                algorithmic_bytes={"images_bf16": 6912 * 42336, "outputs_f32": 6912 * 128, "saved_activations_1536_images": 77800000},
                                how="rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a second pass, --pmc WRITE_SIZE of `python3 bench.py --steps 20 "
                                    "--warmup 3 --no-configs --no-cpu-baseline --no-distribution --no-graph` (scratch/gpu_r04_final.sh), mean over the "
                                    "encoder_fused_kernel<84,84> dispatches; FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, "
                                    "MI355X_MICROARCH.md HBM section).  bench.py reports `traffic` only while encoder_fused_hip_sha256 matches the tree.")}
json.dump(traffic, open(os.path.join(P, "r04_fused_traffic.json"), "w"), indent=1)
marks = open(os.path.join(S, "marks.txt")).read().strip().splitlines()
tl = dict(kv.split("=") for kv in marks[-1].split())
untraced = [l for l in marks if "traced=False" in l][-1].split(":")[-1].strip()
traced = [l for l in marks if "traced=True" in l][-1].split(":")[-1].strip()
with open(os.path.join(P, "r04_step_timeline.md"), "w") as f:
    f.write("# Branch timeline of the headline step, round 4 (device time marks, `scratch/marks2.py`)\n\n"
            "1-thread launches that store the device clock sit between the captured launches of the step; the graph is replayed as usual\n"
            f"(the marks add launches: {traced} traced against {untraced} untraced on the same box - read the intervals).  Microseconds since\n"
            "the first mark of the graph (the eager image pack, ~100 us, runs in front of it); round 3 in brackets (`r03_step_timeline.md`).\n\n"
            "| mark | us | (round 3) |\n|---|---|---|\n")
    r3 = {"front:encoded": 130, "pr:plan": 192, "ad:start": 195, "a:end": 258, "b:q_fwd+cql": 336, "b:actor_bwd": 431, "b:critic_bwd": 449,
          "b:enc_bwd": 700, "ad:end": 731, "c:adam": 732, "step:end": 741}
    for k, v in tl.items():
        f.write(f"| {k} | {v} | {r3.get(k, '')} |\n")
    f.write("\nThe two chains still end together (`c:adam` vs `ad:end`): the main chain's `b:critic_bwd -> b:enc_bwd` segment is "
            f"{int(tl['b:enc_bwd']) - int(tl['b:critic_bwd'])} us (round 3: 251), the action-decoder branch `ad:start -> ad:end` "
            f"{int(tl['ad:end']) - int(tl['ad:start'])} us (536).  What was tried on both this round and why neither moved: DESIGN.md "
            "\"Whole step\", `r04_persistent_stage.md`.\n")
subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "margins_md.py"), os.path.join(ROOT, "gpurun_out", "r4suite", "parity_margins.jsonl")],
               stdout=open(os.path.join(P, "r04_parity_margins.md"), "w"), check=True)
# ---- DESIGN.md placeholders
r, c = line["roofline"], line["configs"]
enc = json.load(open(os.path.join(S, "encoder_launches.json")))
vals = {"ENC_US": f"{r['avg_ms'] * 1e3:.1f}", "ENC_TF": f"{r['achieved']:.0f}", "ENC_FRAC": f"{r['frac'] * 100:.1f}",
        "ENC_ROCPROF_US": f"{enc['in_step_us']['mean']:.1f}", "ENC_N": f"{enc['in_step_us']['n']}", "ENC_OVER_US": f"{r['event_overhead_ms'] * 1e3:.1f}",
        "TRAFFIC_MB": f"{raw['hbm_bytes_per_launch'] / 1e6:.1f}", "TRAFFIC_RD": f"{raw['hbm_read_bytes_corrected'] / 1e6:.1f}",
        "TRAFFIC_WR": f"{raw['hbm_write_bytes'] / 1e6:.1f}", "C3_FRAC": f"{c['c3_tacorl_finetune_b256']['enc_frac'] * 100:.1f}",
        "C5_FRAC": f"{c['c5_cql_n32_b1024']['enc_frac'] * 100:.1f}", "C4_FRAC": f"{c['c4_share_dualcam128_b64']['enc_frac'] * 100:.1f}",
        "STEP_MS": f"{line['ms_per_step']:.4f}", "STEP_VALUE": f"{line['value']:.0f}", "STEP_P50": f"{line['step_time']['median_ms']:.3f}",
        "STEP_P90": f"{line['step_time']['p90_ms']:.3f}", "MARK_ADAM": tl["c:adam"], "MARK_AD": tl["ad:end"],
        "C5_MS": f"{c['c5_cql_n32_b1024']['ms_per_step']:.2f}", "C1B32_MS": f"{c['c1_playlmp_b32']['ms_per_step']:.2f}",
        "C1B256_MS": f"{c['c1_playlmp_b256']['ms_per_step']:.2f}", "C3_MS": f"{c['c3_tacorl_finetune_b256']['ms_per_step']:.2f}",
        "C4_MS": f"{c['c4_share_dualcam128_b64']['ms_per_step']:.2f}", "SEG_INGRAPH": f"{c['c2_three_segment']['one_graph_rccl_nodes']:.3f}",
        "SEG_EAGER": f"{c['c2_three_segment']['three_segments_eager_rccl']:.3f}", "FED_MS": f"{c['c2_fed_from_hbm_replay']['ms_per_step']:.4f}",
        "FED_P50": f"{c['c2_fed_from_hbm_replay']['p50_ms']:.3f}", "FED_P90": f"{c['c2_fed_from_hbm_replay']['p90_ms']:.3f}",
        "FED_P99": f"{c['c2_fed_from_hbm_replay']['p99_ms']:.3f}"}
d = open(os.path.join(ROOT, "DESIGN.md")).read()
missing = set(re.findall(r"\{([A-Z0-9_]+)\}", d)) - set(vals)
assert not missing, missing
for k, v in vals.items():
    d = d.replace("{" + k + "}", v)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(d)
json.dump({"c2_three_segment": c["c2_three_segment"], "c3_strong_share_b32": c["c3_strong_share_b32"],
           "command": "python bench.py (default run; the probe is the child process `python bench.py --probe segments`)",
           "headline_ms_per_step_same_run": line["ms_per_step"]}, open(os.path.join(P, "r04_segment_probe.json"), "w"), indent=1)
print("published; headline", line["ms_per_step"], "ms/step, roofline", r["frac"])
