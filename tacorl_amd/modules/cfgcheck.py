"""Loud validation of the Hydra sub-configs the module classes receive (`_recursive_: False`: they arrive as
dicts, reference config/networks/**).  The HIP path implements the networks the in-scope experiments select
(SURVEY section 2 rows 6-12); every option it does not honour raises instead of training a different model."""


def _leaf(target):
    return str(target).rsplit(".", 1)[-1]


def _require(cfg, what, target_leaf, fixed, free=()):
    """cfg may be empty (defaults).  `_target_` must name `target_leaf`; keys in `fixed` must hold the given value;
    keys in `free` are honoured by the caller; anything else is unknown -> NotImplementedError."""
    cfg = dict(cfg or {})
    t = cfg.pop("_target_", None)
    for k in ("_recursive_", "_convert_", "_partial_"):
        cfg.pop(k, None)
    if t is not None and _leaf(t) != target_leaf:
        raise NotImplementedError(f"{what}: _target_ {t} - only {target_leaf} is on the HIP path")
    for k, v in cfg.items():
        if k in free:
            continue
        if k not in fixed:
            raise NotImplementedError(f"{what}: option {k!r} is not implemented on the HIP path")
        ok = fixed[k]
        if not (v in ok if isinstance(ok, (tuple, list, set)) else v == ok):
            raise NotImplementedError(f"{what}: {k}={v!r} is not implemented on the HIP path (supported: {ok!r})")
    return cfg


ENCODER_FIXED = dict(input_channels=3, latent_dim=32, hidden_dim=256, activation_function="ReLU", dropout=(0, 0.0),
                     temperature=None, normalize_spatial_softmax=False, normalize_output=False, vib=False)


def check_representation(cfg, what, cams=None):
    """config/networks/representation/lmp_encoder.yaml: LateFusion over per-camera LMPVisionEncoders
    (reference representation_network.py:10-71, encoder.py:349-428)."""
    cfg = dict(cfg or {})
    if not cfg:
        return
    nets = cfg.pop("networks", None) or {}
    _require(cfg, what, "LateFusion", {}, free=("modalities",))
    for cam, sub in nets.items():
        if cams is None or cam in cams:
            _require(sub, f"{what}.networks.{cam}", "LMPVisionEncoder", ENCODER_FIXED, free=("device",))
    missing = [c for c in (cams or []) if nets and c not in nets]
    if missing:
        raise ValueError(f"{what}: no encoder configured for modalities {missing}")


def check_goal_encoder(cfg, what, hidden):
    """config/networks/goal_encoder/default.yaml (reference goal_encoder.py:5-33)."""
    _require(cfg, what, "VisualGoalEncoder", dict(hidden_size=hidden, activation_function="ReLU",
                                                  last_layer_activation="Identity"),
             free=("in_features", "out_features"))


def check_actor(cfg, what):
    """config/networks/actor_critic/actor/*.yaml + policy/default.yaml (reference actor.py:18-63,217-270)."""
    cfg = dict(cfg or {})
    pol = cfg.pop("policy", None) or {}
    _require(cfg, what, "Actor", {}, free=("discrete_gripper", "state_dim", "goal_dim", "action_dim"))
    return _require(pol, what + ".policy", "MLPPolicy", {}, free=("num_layers", "hidden_dim"))


def check_critic(cfg, what):
    """config/networks/actor_critic/critic/default.yaml + q_network/default.yaml (reference critic.py:9-30,73-97)."""
    cfg = dict(cfg or {})
    qn = cfg.pop("q_network", None) or {}
    _require(cfg, what, "Critic", {}, free=("state_dim", "goal_dim", "action_dim"))
    return _require(qn, what + ".q_network", "MLPQNetwork", dict(last_layer_activation="Identity"),
                    free=("num_layers", "hidden_dim"))


def check_plan_recognition(cfg, what):
    """config/networks/plan_recognition/transformer.yaml (reference plan_recognition_transformer.py:10-68)."""
    return _require(cfg, what, "PlanRecognitionTransformersNetwork",
                    dict(encoder_normalize=False, positional_normalize=False, position_embedding=True),
                    free=("num_heads", "num_layers", "encoder_hidden_size", "fc_hidden_size", "state_dim",
                          "latent_plan_dim", "min_std", "dropout_p", "max_position_embeddings"))


def check_action_decoder(cfg, what):
    """config/networks/action_decoder/logistic.yaml (reference action_decoder_logistic.py:21-71); the value checks
    (relu rnn_decoder, discrete gripper, +-1 bounds, no dropout) are ActionDecoderLogistic.__init__'s."""
    return _require(cfg, what, "ActionDecoderLogistic", {},
                    free=("n_mixtures", "num_layers", "hidden_size", "out_features", "act_max_bound", "act_min_bound",
                          "policy_rnn_dropout_p", "num_classes", "latent_plan_dim", "rnn_model", "include_goal",
                          "state_dim", "goal_dim", "gripper_alpha", "discrete_gripper"))
