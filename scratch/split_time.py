import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
for split in (False, True):
    mod = bench.build_module(dev, "bf16", 16, 1); batch = bench.synth_batch(256, 16, 84, 84, dev, 1)
    mod._force_graph_split = split
    mod.enable_graph(); mod.log_every_n_steps = 50
    for _ in range(5): mod.training_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): mod.training_step(batch)
    torch.cuda.synchronize()
    print(f"split={split}: {(time.perf_counter() - t0) * 10:.3f} ms/step", flush=True)
