"""Device-time marks of PlayLMP.training_step (graph replay): marks_plmp.py <B>"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib, ops
from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
B = int(sys.argv[1])
actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=16, min_std=1e-4, dropout_p=0.0, max_position_embeddings=16)
ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=16, rnn_model="rnn_decoder", include_goal=False)
cams = ["rgb_static"]
for traced in (False, True):
    reader = ops.trace_marks(dev) if traced else None
    torch.manual_seed(0)
    m = PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams,
                plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16")
    batch = bench.synth_batch(B, 16, 84, 84, dev, 1)
    m.enable_graph(); m.log_every_n_steps = 50
    for _ in range(8): m.training_step(batch, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): m.training_step(batch, 0)
    torch.cuda.synchronize()
    print(f"B={B} traced={traced}: {(time.perf_counter() - t0) * 5:.3f} ms/step", flush=True)
    if traced:
        print("  ".join(f"{n}={t:.0f}" for n, t in reader()))
