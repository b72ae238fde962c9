"""Launch sequence of one TACORL step: (stream, C-ABI entry point) in issue order, with the marks between."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
import tacorl_amd
from tacorl_amd import _lib, ops
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
mod = bench.build_module(dev, "bf16", 16, 1)
batches = [bench.synth_batch(256, 16, 84, 84, dev, 1), bench.synth_batch(256, 16, 84, 84, dev, 2)]
for i in range(3): mod.training_step(batches[i % 2])
torch.cuda.synchronize()
orig = _lib.call
log = []
streams = {}
def traced(name, *a):
    sid = torch.cuda.current_stream().cuda_stream
    streams.setdefault(sid, len(streams))
    log.append((streams[sid], name))
    return orig(name, *a)
for m in list(sys.modules.values()):
    if m and getattr(m, "__name__", "").startswith("tacorl_amd") and getattr(m, "call", None) is orig:
        m.call = traced
mod.training_step(batches[1]); torch.cuda.synchronize()
for s, n in log: print(f"{'    ' * s}[{s}] {n}")
print(len(log), "calls")
