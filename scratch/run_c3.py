import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
from tacorl_amd.modules.tacorl.tacorl import TACORL
cams = ["rgb_static"]; T = 16
actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
critic = {"q_network": {"num_layers": 3, "hidden_dim": 256, "last_layer_activation": "Identity"}}
pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=16, min_std=1e-4, dropout_p=0.0, max_position_embeddings=T)
ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=16, rnn_model="rnn_decoder", include_goal=False)
torch.manual_seed(0)
def lmp():
    return PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams,
                   plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16")
B = int(os.environ.get("B", 256))
batch = bench.synth_batch(B, T, 84, 84, dev, 1)
def timeit(f, steps=int(os.environ.get("STEPS", 20)), warm=int(os.environ.get("WARM", 5))):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
if os.environ.get('ONLY') != 'plmp':
  mod = TACORL(play_lmp=lmp(), finetune_action_decoder=True, critic=critic, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16",
               action_decoder_lr=3e-4, actor_lr=1e-4, critic_lr=3e-4, discount=0.95, conservative_weight=1.0, reward_scale=10.0,
               n_action_samples=4, with_lagrange=True, deterministic_backup=True, bc_epochs=5)
  mod.current_epoch = 5; (None if os.environ.get('NOGRAPH') else mod.enable_graph()); mod.log_every_n_steps = 50
  print(f"C3 TACORL finetune_action_decoder=True B={B}: {timeit(lambda: mod.training_step(batch)):.3f} ms/step", flush=True)
if os.environ.get('ONLY') == 'c3': sys.exit(0)
p = lmp(); p.log_every_n_steps = 50
if os.environ.get('WAVEFRONT') is not None: p.ad.bptt_wavefront = bool(int(os.environ['WAVEFRONT']))
if os.environ.get('BRANCHES') is not None: p.branches = bool(int(os.environ['BRANCHES']))
try:
    (None if os.environ.get('NOGRAPH') else p.enable_graph())
except Exception as e:
    print("playlmp graph:", e)
print(f"PlayLMP.training_step B={B} T={T}: {timeit(lambda: p.training_step(batch, 0)):.3f} ms/step", flush=True)
