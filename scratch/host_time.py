"""Is the headline step host-bound?  Host enqueue time per step (no sync inside the loop) vs device time per step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
mod = bench.build_module(dev, "bf16", 16, 1)
batches = [bench.synth_batch(256, 16, 84, 84, dev, 1), bench.synth_batch(256, 16, 84, 84, dev, 2)]
mod.enable_graph(); mod.log_every_n_steps = 50
for i in range(20): mod.training_step(batches[i % 2])
torch.cuda.synchronize()
for n in (50, 200, 400):
    t0 = time.perf_counter()
    for i in range(n): mod.training_step(batches[i % 2])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e3 * (t1 - t0) / n:.4f} ms/step, total {1e3 * (t2 - t0) / n:.4f} ms/step, drain {1e3 * (t2 - t1):.2f} ms")
# host-only cost of the python around the replay: same loop with the GPU work stubbed is not possible; instead time pieces
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(200): mod.training_step(batches[i % 2])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
