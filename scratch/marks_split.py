"""marks.py for the 3-segment (multi-GPU) graph mode on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib, ops
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
for traced in (False, True):
    reader = ops.trace_marks(dev) if traced else None
    mod = bench.build_module(dev, "bf16", 16, 1)
    batches = [bench.synth_batch(256, 16, 84, 84, dev, 1), bench.synth_batch(256, 16, 84, 84, dev, 2)]
    mod._force_graph_split = True
    mod.enable_graph(); mod.log_every_n_steps = 50
    def run(n):
        for i in range(n): mod.training_step(batches[i % 2])
    run(6)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(100)
    torch.cuda.synchronize()
    print(f"traced={traced}: {(time.perf_counter() - t0) * 10:.3f} ms/step", flush=True)
    if traced:
        for n, t in reader(): print(f"{t:9.1f} us  {n}")
