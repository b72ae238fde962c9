import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import ops, _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for B, cols, reps, ld in [(1024, 64, 97, 72), (1024, 64, 97, 71), (1024, 64, 97, 128), (256, 64, 13, 72), (4096, 64, 24, 72)]:
    ins = [torch.randn(reps * B, ld, device=dev) for _ in range(2)]
    outs = [torch.zeros(B, cols, device=dev) for _ in range(2)]
    f = lambda: ops.call("tacorl_reduce_rows_mod_batch", 2, ops.ptr_array(ins), ld, ops.ptr_array(outs), cols, B, cols, reps, ops.stream())
    t = timeit(f)
    mb = 2 * reps * B * cols * 4 / 1e6
    print(f"B={B} cols={cols} reps={reps} ld={ld}: {t:.1f} us  ({mb:.0f} MB -> {mb / t * 1e6 / 1e6:.2f} TB/s)", flush=True)
