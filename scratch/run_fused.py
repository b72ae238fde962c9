import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacorl_amd import _lib, blocks, ops
if os.environ.get("TACORL_SCRATCH_LIB"): _lib.LIB_PATH = os.environ["TACORL_SCRATCH_LIB"]
dev = torch.device('cuda:0')
H = W = int(os.environ.get("HW", 84))
if os.environ.get("HxW"): H, W = (int(v) for v in os.environ["HxW"].split("x"))
spec = sys.argv[1:] or ["4096", "256", "512", "512"]   # "512a" = that problem also saves its activations (training problems)
n = [int(x.rstrip("a")) for x in spec]
flats, imgs, outs, packed, acts = [], [], [], [], []
for k, sp in zip(n, spec):
    acts.append(torch.zeros(ops.encoder_act_layout(k, H, W)[1], device=dev) if sp.endswith("a") else None)
    flat = torch.randn(blocks.encoder_size(), device=dev) * 0.05
    flats.append(flat)
    imgs.append((torch.rand(k, H, W, 3, device=dev) * 2 - 1).to(torch.bfloat16))
    outs.append(torch.empty(k, 32, device=dev))
    packed.append(torch.empty(_lib.lib().tacorl_encoder_fused_wpk_bytes(), dtype=torch.uint8, device=dev))
ops.call("tacorl_encoder_pack_weights", len(n), ops.ptr_array(flats), ops.ptr_array(packed), ops.stream())
def _flop(H, W):
    o1 = ((H - 8) // 4 + 1, (W - 8) // 4 + 1); o2 = ((o1[0] - 4) // 2 + 1, (o1[1] - 4) // 2 + 1); o3 = (o2[0] - 2, o2[1] - 2)
    return 2.0 * (o1[0] * o1[1] * 32 * 192 + o2[0] * o2[1] * 64 * 512 + o3[0] * o3[1] * 64 * 576 + 128 * 256 + 256 * 32)
FLOP = _flop(H, W)
def run():
    ops.call("tacorl_encoder_fwd_fused", len(n), ops.ptr_array(imgs), ops.ptr_array(packed), ops.ptr_array(flats),
             ops.ptr_array(outs), ops.ptr_array(acts) if any(a is not None for a in acts) else None, ops.int_array(n), H, W, ops.stream())
for _ in range(10): run()
torch.cuda.synchronize()
best = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 50)
ms = sorted(best)[2]
print(f"imgs {sum(n)} ms {ms:.4f} (5 x 50 launches: min {min(best):.4f} max {max(best):.4f}) TF {sum(n)*FLOP/ms/1e9:.1f}")
# phase stamps (only in a -DEF_STAMPS build of the library)
import ctypes as C
L = _lib.lib()
if hasattr(L, "tacorl_ef_stamps_read"):
    buf = (C.c_ulonglong * 64)()
    L.tacorl_ef_stamps_read(buf, 1)
    run(); torch.cuda.synchronize()
    L.tacorl_ef_stamps_read(buf, 0)
    names = ["prologue", "dma issue", "conv1", "barrier1", "conv2", "barrier2", "conv3 mfma", "soft-argmax", "fc tail", "img wait+barrier"]
    if os.environ.get("EF8_NAMES"):
        names = ["prologue", "conv1", "B1", "conv2 chains+store", "B2", "conv2 finish", "B3", "conv3 chains+store", "B4", "conv3 finish+softargmax", "B5(+land)", "merge+FC+B6", ""]
    if any(buf[10:13]): print("band phases: compute", buf[10], "own DMA wait", buf[11], "barrier", buf[12])
    tot = sum(buf[:13])
    for k, nm in enumerate(names): print(f"{nm:18s} {buf[k]:9d} clk  {100.0 * buf[k] / tot:5.1f} %")
    print("total clk", tot)
    if any(buf[16:64]):
        print("fine stamps (conv1 tiles 16.., conv2 [blocks12, blocks34, epilogue] x tile 24..):")
        print("  conv1", [buf[k] for k in range(16, 24)])
        print("  conv2", [buf[k] for k in range(24, 33)])

if hasattr(L, "tacorl_ef_blk_read"):
    blk = (C.c_ulonglong * 512)()
    run(); torch.cuda.synchronize()
    L.tacorl_ef_blk_read(blk)
    import collections
    by = collections.defaultdict(list)
    for b in range(256):
        if blk[256 + b]: by[int(blk[256 + b])].append(int(blk[b]))
    print("workgroup clocks by images served:", {k: (len(v), min(v), sum(v) // len(v), max(v)) for k, v in sorted(by.items())}, "(count, min, mean, max)")
    print("critical workgroup:", max(int(blk[b]) for b in range(256)), "clk; mean", sum(int(blk[b]) for b in range(256)) // 256)
