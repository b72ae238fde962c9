"""Timing experiment (results of the side pass are NOT meaningful here): the logging-only action-decoder pass taken out of
the step's graph and replayed as its own graph on a side stream at the START of the next step (beside the image pack, the
encoder forward and the policy / Q phases) instead of as a branch from the plan to the end of the step.
Variants: base = the product step; defer = side graph launched at step start; defer2 = two halves, the second one released
behind the encoder forward (approximated: launched after the main graph's launch call returns... not gated)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
B, T = 256, 16
batches = [bench.synth_batch(B, T, 84, 84, dev, 1), bench.synth_batch(B, T, 84, 84, dev, 2)]

def timeit(step, n=400, rounds=3):
    for _ in range(20): step(0)
    torch.cuda.synchronize()
    out = []
    for r in range(rounds):
        t0 = time.perf_counter()
        for i in range(n): step(i)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    return out

base = bench.build_module(dev, "bf16", T, 1)
base.enable_graph(); base.log_every_n_steps = 50
print("base     ", " ".join(f"{x:.4f}" for x in timeit(lambda i: base.training_step(batches[i % 2]))), flush=True)
del base
import gc; gc.collect(); torch.cuda.empty_cache()

mod = bench.build_module(dev, "bf16", T, 1, ad_every=10 ** 9)
mod.enable_graph(); mod.log_every_n_steps = 50
for i in range(5): mod.training_step(batches[i % 2])
print("no AD    ", " ".join(f"{x:.4f}" for x in timeit(lambda i: mod.training_step(batches[i % 2]))), flush=True)
side = torch.cuda.Stream(device=dev)
def ad_pass():
    mod.ad.loss_step(mod, mod.acts, mod.plan, B, T, False, frozen=True)
ad_pass(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    ad_pass()
torch.cuda.synchronize()
def step_defer(i):
    with torch.cuda.stream(side):
        g.replay()               # the previous step's pass (its inputs would be a private copy in the product)
    mod.training_step(batches[i % 2])
print("defer    ", " ".join(f"{x:.4f}" for x in timeit(step_defer)), flush=True)
ev = torch.cuda.Event()
def step_defer_after_pack(i):
    # released once the eager image pack is queued: beside the encoder forward and everything after it
    mod._stage_hook = None
    mod.training_step(batches[i % 2])
    with torch.cuda.stream(side):
        g.replay()
print("defer-end", " ".join(f"{x:.4f}" for x in timeit(step_defer_after_pack)), flush=True)
