"""A/B of a module attribute on the headline step, alternating on one box: ab_step.py attr v0 v1 [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
attr, v0, v1 = sys.argv[1], eval(sys.argv[2]), eval(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
BSZ = int(os.environ.get("BSZ", 256))  # FINETUNE=1: C3 (action-decoder fine-tuning); BSZ: batch size
mod = bench.build_module(dev, "bf16", 16, 1, finetune=bool(int(os.environ.get("FINETUNE", 0))))
batches = [bench.synth_batch(BSZ, 16, 84, 84, dev, 1), bench.synth_batch(BSZ, 16, 84, 84, dev, 2)]
mod.enable_graph(); mod.log_every_n_steps = 50
def setv(path, v):
    torch.cuda.synchronize()  # (no graph may still be replaying when the captures are dropped)
    if path.startswith("env:"):
        os.environ[path[4:]] = str(v)
        mod._graphs = {}
        return
    o = mod
    if path.startswith("ops:"):  # a switch of tacorl_amd.ops, e.g. ops:prep_batch.enabled
        from tacorl_amd import ops as o
        path = path[4:]
    *head, last = path.split(".")
    for h in head: o = getattr(o, h)
    setattr(o, last, v)
    if hasattr(mod.engine, "_lean_cache"): mod.engine._lean_cache = {}
    if hasattr(mod.engine, "_gather_cache"): mod.engine._gather_cache = {}
    mod._graphs = {}  # the captured step depends on the switch
def run(n):
    for i in range(n): mod.training_step(batches[i % 2])
for v in (v0, v1):
    setv(attr, v); run(10)
torch.cuda.synchronize()
res = {repr(v0): [], repr(v1): []}
for r in range(rounds):
    for v in (v0, v1):
        setv(attr, v); run(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(400); torch.cuda.synchronize()
        res[repr(v)].append((time.perf_counter() - t0) / 400 * 1e3)
for k, v in res.items():
    print(f"{attr}={k}: " + " ".join(f"{x:.4f}" for x in v) + f"  mean {sum(v) / len(v):.4f} ms/step")
