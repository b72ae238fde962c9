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
