"""Timing probe (results racy on purpose): the eager image pack issued on its own stream so that it overlaps the step's graph
instead of running in front of it - what pipelining the pack of step n+1 beside step n would return."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
mod = bench.build_module(dev, "bf16", 16, 1)
batches = [bench.synth_batch(256, 16, 84, 84, dev, 1), bench.synth_batch(256, 16, 84, 84, dev, 2)]
mod.enable_graph(); mod.log_every_n_steps = 50
orig = mod._stage_frames
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
streams = {"normal": torch.cuda.Stream(device=dev), "low": torch.cuda.Stream(device=dev, priority=lo)}
mode = [None]
last = [None]
def staged(batch, noise, nchw=True):
    if mode[0] == "skip":
        return last[0]
    if mode[0] is None:
        last[0] = orig(batch, noise, nchw)
        return last[0]
    if mode[0] is None:
        return orig(batch, noise, nchw)
    with torch.cuda.stream(streams[mode[0]]):
        return orig(batch, noise, nchw)
mod._stage_frames = staged
def run(n):
    for i in range(n): mod.training_step(batches[i % 2])
run(10); torch.cuda.synchronize()
res = {}
for r in range(3):
    for m in (None, "normal"):
        mode[0] = m
        run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(400); th = time.perf_counter() - t0; torch.cuda.synchronize()
        res.setdefault(m, []).append((time.perf_counter() - t0) / 400 * 1e3)
        res.setdefault(str(m) + " host-enqueue", []).append(th / 400 * 1e3)
for k, v in res.items():
    print(f"pack stream={k}: " + " ".join(f"{x:.4f}" for x in v) + f"  mean {sum(v) / len(v):.4f} ms/step")
