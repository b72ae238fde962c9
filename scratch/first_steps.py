"""Per-step times of the first steps behind the driver's 5 warm-up steps (why does --steps 20 read 2 % above the 200-step mean?)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
mod = bench.build_module(dev, "bf16", 16, 1)
batch = bench.synth_batch(256, 16, 84, 84, dev, 1)
mod.enable_graph(); mod.log_every_n_steps = 50
for i in range(2): mod.training_step(batch)
torch.cuda.synchronize()
pre = int(os.environ.get("PRE_MS", 0))
if pre:  # chip conditioning: the step's encoder launch (no parameter changes) back to back for PRE_MS
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < pre:
        for _ in range(20): mod.engine.encode_fused_only()
        torch.cuda.synchronize()
for i in range(3): mod.training_step(batch)
torch.cuda.synchronize()
N = 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(N):
    mod.training_step(batch); ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print("wall per step over the first 20: %.4f  over 120: %.4f" % (sum(ms[:20]) / 20, sum(ms) / N))
print(" ".join(f"{x:.3f}" for x in ms[:40]))
print(" ".join(f"{x:.3f}" for x in ms[40:80]))
