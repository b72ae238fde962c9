"""marks.py with engine switches: usage marks2.py attr=value ... (class attributes of ACEngine)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib, ops
from tacorl_amd.engine import ACEngine
for kv in sys.argv[1:]:
    k, v = kv.split("="); setattr(ACEngine, k, eval(v))
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
for traced in (False, True):
    reader = ops.trace_marks(dev) if traced else None
    mod = bench.build_module(dev, "bf16", 16, 1, finetune=bool(int(os.environ.get("FINETUNE", 0))))
    batches = [bench.synth_batch(256, 16, 84, 84, dev, 1), bench.synth_batch(256, 16, 84, 84, dev, 2)]
    mod.enable_graph(); mod.log_every_n_steps = 50
    def run(n):
        for i in range(n): mod.training_step(batches[i % 2])
    run(6); torch.cuda.synchronize(); t0 = time.perf_counter(); run(200); torch.cuda.synchronize()
    print(f"{sys.argv[1:]} traced={traced}: {(time.perf_counter() - t0) * 5:.3f} ms/step", flush=True)
    if traced:
        print("  ".join(f"{n}={t:.0f}" for n, t in reader()))
