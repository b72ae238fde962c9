"""PlayLMP bf16: validation_step (eager and graph) next to training_step - same logs from the same batch/noise tape"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
from tacorl_amd import _lib
from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
actor = {"policy": {"num_layers": 3, "hidden_dim": 256}}
pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=16, min_std=1e-4, dropout_p=0.0, max_position_embeddings=16)
ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=16, rnn_model="rnn_decoder", include_goal=False)
cams = ["rgb_static"]
res = {}
for graph in (False, True):
    torch.manual_seed(0)
    m = PlayLMP(plan_proposal=actor, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams,
                plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16")
    batch = bench.synth_batch(32, 16, 84, 84, dev, 1)
    if graph: m.enable_graph()
    m.eval()
    torch.manual_seed(3); torch.cuda.manual_seed(3)
    for _ in range(2): m.validation_step(batch, 0)
    torch.cuda.synchronize()
    res[graph] = dict(m.logged)
print({k: round(v, 6) for k, v in res[False].items()})
assert res[False].keys() == res[True].keys() and all(v == v for v in res[True].values())
print("validation eager/graph keys ok; max rel diff", max(abs(res[False][k] - res[True][k]) / max(1e-9, abs(res[False][k])) for k in res[False]))
