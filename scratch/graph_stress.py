"""How many live instantiated hipGraphs does this stack take?  N small TACORL modules (B = 8), every one with its captured
step graph alive, replayed round-robin.  usage: graph_stress.py N [rounds]   (run each N in a process of its own: a
segfault in hipGraphLaunch ends the process)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
N, rounds = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
mods, batches = [], []
for i in range(N):
    m = bench.build_module(dev, "bf16", 16, 1)
    m.enable_graph(); m.log_every_n_steps = 10 ** 9
    b = bench.synth_batch(8, 16, 84, 84, dev, i)
    for _ in range(3): m.training_step(b)   # eager pass, capture, first replay
    torch.cuda.synchronize()
    mods.append(m); batches.append(b)
    print(f"captured {i + 1}", flush=True)
t0 = time.perf_counter()
for r in range(rounds):
    for m, b in zip(mods, batches): m.training_step(b)
torch.cuda.synchronize()
print(f"N={N}: {rounds} rounds of replays ok, {(time.perf_counter() - t0) / (rounds * N) * 1e3:.3f} ms/step; graphs alive: {sum(len(m._graphs) for m in mods)}")
