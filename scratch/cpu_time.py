import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
mod = bench.build_module(dev, "bf16", 16, 1); batch = bench.synth_batch(256, 16, 84, 84, dev, 1)
mod.enable_graph(); mod.log_every_n_steps = 50
for _ in range(5): mod.training_step(batch)
torch.cuda.synchronize()
# CPU issue time: enqueue 20 steps without waiting for the GPU
t0 = time.perf_counter()
for _ in range(20): mod.training_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"cpu enqueue per step {1e3*(t1-t0)/20:.3f} ms; total per step {1e3*(t2-t0)/20:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): mod.training_step(batch)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
