"""The replay-fed step (HbmReplay.batch(fused=True) behind prefetching()) against the resident batch, with different GIL
switch intervals and prefetch depths: where the feeder's +4 % comes from."""
import os, sys, time, argparse
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
from tacorl_amd import _lib
from tacorl_amd.data.replay import HbmReplay, PlayIndex, prefetching
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
B, T, H, W = 256, 16, 84, 84
mod = bench.build_module(dev, "bf16", T, 1)
mod.enable_graph(); mod.log_every_n_steps = 50
batch = bench.synth_batch(B, T, H, W, dev, 1)
n_frames = 40000
g = torch.Generator().manual_seed(5)
frames = torch.randint(0, 256, (n_frames, H, W, 3), dtype=torch.uint8, generator=g)
acts = np.random.RandomState(6).uniform(-1, 1, size=(n_frames, 7)).astype(np.float32)
ix = PlayIndex([[i, i + 1999] for i in range(0, n_frames, 2000)], T, T, goal_sampling_prob=0.3)
rng = np.random.default_rng(7)
rep = HbmReplay({"rgb_static": frames}, acts, ix, dev)
def draw():
    return rng.integers(len(ix), size=B), ix.draw(B, rng), None
def make():
    return rep.batch(*draw(), fused=True)
def timed(fn, n=600):
    fn(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def resident(n):
    for _ in range(n): mod.training_step(batch)
def fed(n, depth=2):
    for b in prefetching(make, n, depth=depth): mod.training_step(b)
def inline(n):
    for _ in range(n): mod.training_step(make())
import argparse
a = argparse.Namespace(feeder="hbm", augment=False, steps=200, warmup=10)
barrier = torch.cuda.synchronize
for _ in range(30): mod.training_step(batch)
fed(10); torch.cuda.synchronize()
import queue, threading
q = queue.Queue(maxsize=2)
prod = []
def produce(n):
    torch.cuda.set_device(0)
    for _ in range(n):
        t0 = time.perf_counter()
        b = make()
        t1 = time.perf_counter()
        q.put(b)
        prod.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
th = threading.Thread(target=produce, args=(400,), daemon=True); th.start()
wait, step = [], []
import faulthandler
fh = open("/tmp/stalls.txt", "w")
for i in range(400):
    t0 = time.perf_counter(); b = q.get(); t1 = time.perf_counter()
    if i % 50 != 9: faulthandler.dump_traceback_later(0.006, file=fh)
    mod.training_step(b); t2 = time.perf_counter()
    faulthandler.cancel_dump_traceback_later()
    wait.append((t1 - t0) * 1e3); step.append((t2 - t1) * 1e3)
torch.cuda.synchronize(); th.join()
big = lambda xs, lim: [(i, round(x, 1)) for i, x in enumerate(xs) if x > lim][:12]
print("main: waits on the queue > 1 ms:", big(wait, 1.0))
print("main: training_step host time > 2 ms:", big(step, 2.0))
print("producer: make_batch > 1 ms:", big([p[0] for p in prod], 1.0))
fh.close(); print(open("/tmp/stalls.txt").read()[:6000])
print("means: wait %.3f step %.3f make %.3f" % (sum(wait) / 400, sum(step) / 400, sum(p[0] for p in prod) / 400))
sys.exit(0)
for _ in range(30): mod.training_step(batch)
print("time_feeder first:", bench.time_feeder(mod, a, B, T, H, W, dev, barrier, lambda x: x)["ms_per_step"])
print("dist:", bench.step_time_distribution(mod, batch, 0.88)["median_ms"])
print("time_feeder after dist:", bench.time_feeder(mod, a, B, T, H, W, dev, barrier, lambda x: x)["ms_per_step"])
print("enc fwd probe:", bench.time_encoder_fwd(mod, B, H, W))
print("time_feeder after enc fwd probe:", bench.time_feeder(mod, a, B, T, H, W, dev, barrier, lambda x: x)["ms_per_step"])
print("in-step probe:", bench.time_encoder_in_step(mod, batch))
print("time_feeder after in-step probe:", bench.time_feeder(mod, a, B, T, H, W, dev, barrier, lambda x: x)["ms_per_step"])
sys.exit(0)
t0 = time.perf_counter()
for _ in range(200): make()
torch.cuda.synchronize()
print(f"host side of one batch (make_batch alone): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms")
print(f"resident fp32 batch : {timed(resident):.4f} ms/step")
u8 = make()
def resident_u8(n):
    for _ in range(n): mod.training_step(u8)
print(f"resident replay batch (same index tables every step): {timed(resident_u8):.4f} ms/step")
print(f"inline (no thread)  : {timed(inline):.4f}")
for si in (5e-3, 1e-3, 2e-4, 5e-5):
    sys.setswitchinterval(si)
    print(f"switch interval {si:g}: depth 2 {timed(lambda n: fed(n, 2)):.4f}  depth 4 {timed(lambda n: fed(n, 4)):.4f}")
sys.setswitchinterval(5e-3)
for n in (100, 200, 400, 1600):
    fed(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); fed(n); torch.cuda.synchronize()
    print(f"fed, {n} steps: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step")
import argparse
a = argparse.Namespace(feeder="hbm", augment=False, steps=200, warmup=10)
barrier = torch.cuda.synchronize
print("bench.time_feeder:", bench.time_feeder(mod, a, B, T, H, W, dev, barrier, lambda x: x)["ms_per_step"])
a.steps = 800
print("bench.time_feeder 800 steps:", bench.time_feeder(mod, a, B, T, H, W, dev, barrier, lambda x: x)["ms_per_step"])
