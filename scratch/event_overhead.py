"""What a HIP-event bracket adds to a kernel's duration on an idle vs busy stream (1-thread time-mark kernel inside)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacorl_amd import ops
from tacorl_amd._lib import call, ptr
dev = torch.device("cuda:0")
buf = torch.zeros(8, dtype=torch.int64, device=dev)
x = torch.randn(4096, 4096, device=dev)
def bracket(busy):
    ms = []
    for _ in range(200):
        if busy:
            y = x @ x  # something ahead in the stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call("tacorl_time_mark", ptr(buf), 0, ops.stream()); e1.record()
        ms.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ms)
    return v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]
print("idle stream: median %.2f us (p10 %.2f p90 %.2f)" % bracket(False))
print("busy stream: median %.2f us (p10 %.2f p90 %.2f)" % bracket(True))
