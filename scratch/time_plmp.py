"""PlayLMP.training_step ms/step (hipGraph), B from env: same-box A/B with TACORL_HIP_LIB=..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
cams = ["rgb_static"]; T = 16
actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=16, min_std=1e-4, dropout_p=0.0, max_position_embeddings=T)
ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=16, rnn_model="rnn_decoder", include_goal=False)
torch.manual_seed(0)
p = PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams,
            plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16")
B = int(os.environ.get("B", 256))
batch = bench.synth_batch(B, T, 84, 84, dev, 1)
p.enable_graph(); p.log_every_n_steps = 50
for _ in range(10): p.training_step(batch, 0)
torch.cuda.synchronize()
res = []
for r in range(3):
    t0 = time.perf_counter()
    for _ in range(200): p.training_step(batch, 0)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 200 * 1e3)
print(f"PlayLMP B={B}: " + " ".join(f"{x:.4f}" for x in res))
