"""Module configs as Hydra would hand them to `instantiate` after composing the reference's YAMLs
(config/module/{cql_offline_goal_cond,tacorl,play_lmp_for_rl}.yaml with their defaults lists resolved;
`_recursive_: False`, so sub-configs stay dicts).  `_target_`s of the sub-configs keep the reference's dotted paths -
the HIP modules check the leaf class name; the module `_target_` is the tacorl_amd replacement."""
ENC = {"_target_": "tacorl.networks.visual_encoders.encoder.LMPVisionEncoder", "latent_dim": 32, "hidden_dim": 256,
       "normalize_output": False}


def representation(cams):
    return {"_target_": "tacorl.networks.representation.representation_network.LateFusion", "_recursive_": False,
            "networks": {c: dict(ENC) for c in cams}}


GOAL_ENC = {"_target_": "tacorl.networks.visual_encoders.goal_encoder.VisualGoalEncoder", "in_features": None,
            "out_features": None, "activation_function": "ReLU", "last_layer_activation": "Identity", "hidden_size": 256}
POLICY = {"_target_": "tacorl.networks.actor_critic.actor.MLPPolicy", "num_layers": 3, "hidden_dim": 256}
ACTOR = {"_target_": "tacorl.networks.actor_critic.actor.Actor", "_recursive_": False, "policy": POLICY}
CRITIC = {"_target_": "tacorl.networks.actor_critic.critic.Critic", "_recursive_": False,
          "q_network": {"_target_": "tacorl.networks.actor_critic.critic.MLPQNetwork", "num_layers": 3, "hidden_dim": 256,
                        "last_layer_activation": "Identity"}}


def plan_recognition(latent, T, dropout_p=0.0):
    return {"_target_": "tacorl.networks.plan_encoders.plan_recognition_transformer.PlanRecognitionTransformersNetwork",
            "num_heads": 8, "num_layers": 2, "encoder_hidden_size": 2048, "fc_hidden_size": 4096, "state_dim": None,
            "latent_plan_dim": latent, "min_std": 0.0001, "dropout_p": dropout_p, "encoder_normalize": False,
            "positional_normalize": False, "position_embedding": True, "max_position_embeddings": T}


def action_decoder(latent):
    return {"_target_": "tacorl.networks.action_decoders.action_decoder_logistic.ActionDecoderLogistic", "n_mixtures": 10,
            "num_layers": 2, "hidden_size": 2048, "out_features": 7, "act_max_bound": [1.0] * 7,
            "act_min_bound": [-1.0] * 7, "policy_rnn_dropout_p": 0.0, "num_classes": 10, "latent_plan_dim": latent,
            "rnn_model": "rnn_decoder", "include_goal": False}


def cql_cfg(cams=("rgb_static",), device="cpu", **over):
    cams = list(cams)
    cfg = {"_target_": "tacorl_amd.modules.cql.cql_offline_lightning.CQL_Offline", "_recursive_": False,
           "actor": dict(ACTOR, discrete_gripper=True), "critic": CRITIC, "actor_encoder": representation(cams),
           "critic_encoder": representation(cams), "goal_encoder": GOAL_ENC,
           # config/module/cql_offline_goal_cond.yaml:11-27
           "discount": 0.99, "actor_lr": 1e-4, "critic_lr": 3e-4, "conservative_weight": 1.0, "n_action_samples": 4,
           "with_lagrange": True, "reward_scale": 10.0, "deterministic_backup": False, "bc_epochs": 5,
           "real_world": True, "obs_modalities": cams, "goal_modalities": cams, "action_dim": 7, "device": device}
    cfg.update(over)
    return cfg


def playlmp_cfg(cams=("rgb_static",), latent=16, T=16, device="cpu", dropout_p=0.0, **over):
    cams = list(cams)
    cfg = {"_target_": "tacorl_amd.modules.play_lmp.play_lmp_for_rl.PlayLMP", "_recursive_": False,
           "plan_proposal": ACTOR, "plan_recognition": plan_recognition(latent, T, dropout_p),
           "goal_encoder": GOAL_ENC, "perceptual_encoder": representation(cams), "action_decoder": action_decoder(latent),
           "lr": 1e-4, "kl_beta": 1e-3, "plan_proposal_obs_modalities": cams, "plan_proposal_goal_modalities": cams,
           "plan_recognition_modalities": cams, "action_decoder_modalities": cams, "real_world": True, "device": device}
    cfg.update(over)
    return cfg


def tacorl_cfg(cams=("rgb_static",), device="cpu", **over):
    cfg = {"_target_": "tacorl_amd.modules.tacorl.tacorl.TACORL", "_recursive_": False, "critic": CRITIC,
           "critic_encoder": representation(list(cams)),
           # config/module/tacorl.yaml:8-30
           "finetune_action_decoder": True, "action_decoder_lr": 3e-4, "actor_lr": 1e-4, "critic_lr": 3e-4,
           "discount": 0.95, "conservative_weight": 1.0, "reward_scale": 10.0, "n_action_samples": 4,
           "with_lagrange": True, "deterministic_backup": True, "bc_epochs": 5, "with_dr3": False,
           "dr3_coefficient": 0.03, "with_vib": False, "vib_coefficient": 0.03, "real_world": True, "device": device}
    cfg.update(over)
    return cfg


def write_reference_run_dir(root, lmp_state_dict, latent=16, T=16, cams=("rgb_static",), extra_ckpts=()):
    """A PlayLMP run directory as the reference's training leaves it (Hydra's `.hydra/config.yaml` with its
    interpolations unresolved - `${latent_plan_dim}`, `${datamodule.dataset.max_window_size}`,
    config/networks/plan_recognition/transformer.yaml:7,13 - and PL checkpoints under model_ckpts/): what
    `load_pl_module_from_checkpoint` (utils/networks.py:90-142) and TACORL(play_lmp_dir=...) consume."""
    import os

    import torch
    import yaml

    cams = list(cams)
    pr = dict(plan_recognition(latent, T), latent_plan_dim="${latent_plan_dim}",
              max_position_embeddings="${datamodule.dataset.max_window_size}")
    ad = dict(action_decoder(latent), latent_plan_dim="${latent_plan_dim}")
    cfg = {
        "latent_plan_dim": latent, "seed": 42,
        "datamodule": {"dataset": {"max_window_size": T, "min_window_size": T}},
        "module": {"_target_": "tacorl.modules.play_lmp.play_lmp_for_rl.PlayLMP", "_recursive_": False,
                   "plan_proposal": ACTOR, "plan_recognition": pr, "goal_encoder": GOAL_ENC,
                   "perceptual_encoder": representation(cams), "action_decoder": ad, "lr": 1e-4, "kl_beta": 1e-3,
                   "plan_proposal_obs_modalities": cams, "plan_proposal_goal_modalities": cams,
                   "plan_recognition_modalities": cams, "action_decoder_modalities": cams, "real_world": True},
    }
    os.makedirs(os.path.join(root, ".hydra"), exist_ok=True)
    os.makedirs(os.path.join(root, "model_ckpts"), exist_ok=True)
    with open(os.path.join(root, ".hydra", "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    ck = {"epoch": 3, "global_step": 30, "pytorch-lightning_version": "1.6.5", "state_dict": dict(lmp_state_dict),
          "hyper_parameters": {"lr": 1e-4}}
    torch.save(ck, os.path.join(root, "model_ckpts", "last.ckpt"))
    for name, sd in extra_ckpts:
        torch.save(dict(ck, state_dict=dict(sd)), os.path.join(root, "model_ckpts", name))
    return root


def lmp_state_dict_from_tacorl(params):
    """PlayLMP state-dict keys (SURVEY 8a note 9) out of a TACORL parameter dict: the LMP's frozen pieces plus the
    actor's head / goal encoder standing in for plan_proposal / goal_encoder."""
    sd = {}
    for k, v in params.items():
        if k.startswith(("perceptual_encoder.", "plan_recognition.", "action_decoder.")):
            sd[k] = v
        elif k.startswith("actor.actor.policy."):
            sd["plan_proposal.policy." + k[len("actor.actor.policy."):]] = v
        elif k.startswith("actor.goal_encoder."):
            sd[k[len("actor."):]] = v
    import torch

    # buffers the reference registers (action_decoder_logistic.py:60-62, action_decoder.py:22-40): part of its checkpoints
    sd.update({"action_decoder.one_hot_embedding_eye": torch.eye(10), "action_decoder.ones": torch.ones(1, 1, 10),
               "action_decoder.gripper_bounds": torch.tensor([-1.0, 1.0]),
               "action_decoder.action_max_bound": torch.ones(1, 1, 6, 10),
               "action_decoder.action_min_bound": -torch.ones(1, 1, 6, 10)})
    return sd
