"""Time the single-launch plan-recognition inference (with / without the in-launch head + sample) at B=256."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import _lib, ops
from tacorl_amd.networks.plan_recognition import PlanRecognition
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
B, T, D, A = [int(x) for x in os.environ.get("PR_SHAPE", "256,16,32,16").split(",")]
pr = PlanRecognition(state_dim=D, latent_plan_dim=A, device=dev, num_heads=8, num_layers=2, encoder_hidden_size=2048,
                     fc_hidden_size=4096, max_position_embeddings=T, trainable=False)
emb = torch.randn(B * T, D, device=dev); eps = torch.randn(B, A, device=dev); plan = torch.zeros(B, A, device=dev)
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
pr.prepare_inference()
print(f"prepare_inference (bf16 mirror + head compose): {timeit(pr.prepare_inference):.1f} us")
print(f"fused encoder + 2 GEMM head:                    {timeit(lambda: pr.forward(emb, D, B, T, 1, inference=True, prepared=True)):.1f} us")
print(f"fused encoder + head + sample in launch:        {timeit(lambda: pr.forward(emb, D, B, T, 1, inference=True, sample=(eps, plan), prepared=True)):.1f} us")
# phase stamps (only in a -DPR_STAMPS build: scratch/mklib_file.sh pr_st pr_fused.hip -DPR_STAMPS)
import ctypes as C
L = _lib.lib()
if hasattr(L, "tacorl_pr_stamps_read"):
    buf = (C.c_ulonglong * 16)()
    f = lambda: pr.forward(emb, D, B, T, 1, inference=True, sample=(eps, plan), prepared=True)
    f(); torch.cuda.synchronize()
    L.tacorl_pr_stamps_read(buf, 1)
    for _ in range(10): f()
    torch.cuda.synchronize()
    L.tacorl_pr_stamps_read(buf, 0)
    names = ["layer top (operand requests; first: x load)", "q|k|v", "attention", "out-proj + LN1 (+ put_xb)", "FFN chunks", "exchange + LN2"]
    tot = sum(buf[:6])
    for k, nm in enumerate(names): print(f"{nm:46s} {buf[k] / 10:9.0f} clk  {100.0 * buf[k] / tot:5.1f} %")
    print("total clk per launch (wave 0 of workgroup 0, both layers)", tot / 10)
