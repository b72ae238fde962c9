"""Full-size sanity/timing of the non-headline BASELINE configs on one GPU:
C5 (CQL_Offline, discrete gripper, A=7, n=32, B=1024, 84x84) and C4-like (TACORL dual camera 128x128, A=32, T=32, B=64/GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import synth, _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
ACTOR = {"policy": {"num_layers": 3, "hidden_dim": 256}}
CRITIC = {"q_network": {"num_layers": 3, "hidden_dim": 256, "last_layer_activation": "Identity"}}
def to_dev(x):
    if isinstance(x, dict): return {k: to_dev(v) for k, v in x.items()}
    return x.to(dev) if torch.is_tensor(x) else x
def timeit(mod, batch, args=(), steps=20, warm=5):
    for _ in range(warm): mod.training_step(batch, *args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): mod.training_step(batch, *args)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
if which == "c5":
    from tacorl_amd.modules.cql.cql_offline_lightning import CQL_Offline
    B = int(os.environ.get("B", 1024))
    mod = CQL_Offline(actor=dict(ACTOR, discrete_gripper=True), critic=CRITIC, real_world=True, obs_modalities=["rgb_static"],
                      goal_modalities=["rgb_static"], action_dim=7, device="cuda:0", compute_dtype="bf16", image_dtype="bf16",
                      discount=0.99, actor_lr=1e-4, critic_lr=3e-4, conservative_weight=1.0, n_action_samples=32,
                      with_lagrange=True, reward_scale=10.0, deterministic_backup=False, bc_epochs=5)
    mod.current_epoch = 5
    batch = to_dev(synth.make_transition_batch(7, B, {"rgb_static": (84, 84)}))
    (None if os.environ.get('NOGRAPH') else mod.enable_graph()); mod.log_every_n_steps = 50
    ms = timeit(mod, batch, (0,))
    logs = mod.engine.metrics()
    print(f"C5 CQL_Offline B={B} n=32 bf16: {ms:.3f} ms/step = {B / ms * 1e3:.0f} samples/s; finite={all(v == v for v in logs.values())}")
else:
    import bench
    from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
    from tacorl_amd.modules.tacorl.tacorl import TACORL
    B, T, A = int(os.environ.get("B", 64)), 32, 32
    cams = ["rgb_gripper", "rgb_static"]
    pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=A, min_std=1e-4, dropout_p=0.0, max_position_embeddings=T)
    ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=A, rnn_model="rnn_decoder", include_goal=False)
    torch.manual_seed(0)
    lmp = PlayLMP(plan_proposal=ACTOR, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams,
                  plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16")
    mod = TACORL(play_lmp=lmp, finetune_action_decoder=False, critic=CRITIC, real_world=True, device=dev, compute_dtype="bf16", image_dtype="bf16",
                 action_decoder_lr=3e-4, actor_lr=1e-4, critic_lr=3e-4, discount=0.95, conservative_weight=1.0, reward_scale=10.0,
                 n_action_samples=4, with_lagrange=True, deterministic_backup=True, bc_epochs=5)
    mod.current_epoch = 5
    g = torch.Generator(device=dev).manual_seed(3)
    u = lambda *s: torch.rand(*s, device=dev, generator=g) * 2 - 1
    acts = u(B, T, 7); acts[..., -1] = torch.where(acts[..., -1] >= 0, 1.0, -1.0)
    disp = torch.ones(B, device=dev).long()
    hw = {"rgb_gripper": (84, 84), "rgb_static": (150, 200)} if which == "c4real" else {c: (128, 128) for c in cams}  # c4real: experiment=tacorl_real_world's geometry
    batch = {"states": {c: u(B, T, 3, *hw[c]) for c in cams}, "goal": {c: u(B, 3, *hw[c]) for c in cams}, "actions": acts, "disp": disp}
    (None if os.environ.get('NOGRAPH') else mod.enable_graph()); mod.log_every_n_steps = 50
    if os.environ.get("MARKS"):  # device-mark timeline of the step's branches (MARKS=1; the module must be built after this)
        from tacorl_amd import ops
        reader = ops.trace_marks(dev)
        mod._graphs = {}
        timeit(mod, batch, steps=10, warm=3)
        print("  ".join(f"{n}={t:.0f}" for n, t in reader()))
    ms = timeit(mod, batch, steps=10, warm=3)
    logs = mod.engine.metrics()
    print(f"C4-like TACORL dual-cam 128x128 A=32 T=32 B={B} bf16: {ms:.3f} ms/step = {B / ms * 1e3:.0f} samples/s; finite={all(v == v for v in logs.values())}")
