"""Reference-import harness (THIS CONTAINER ONLY; test infrastructure, never shipped).

Imports the *unmodified* reference sources from /root/reference on CPU by
registering small stand-ins for the third-party packages that are absent here
(pytorch_lightning, hydra, omegaconf, gym, cv2, torchvision, ...).  Recipe:
SURVEY.md Appendix A.  Used by oracle/gen_golden.py to produce the fixtures in
tests/golden/ and by nothing else.  Nothing in tests/ -m gpu, bench.py or
__graft_entry__.smoke() imports this file (the GPU box has no /root/reference).

Noise capture: torch.distributions.Normal.{sample,rsample}, Tensor.uniform_ and
torch.rand are wrapped so every random draw made by the reference during a step
is (a) taken from an explicit standard tensor that is recorded in order and
(b) applied as `loc + eps * scale` (the op order torch itself uses).  A golden
is therefore "reference code + injected noise", which the restatement and the
HIP engine consume as explicit inputs.
"""
import copy
import importlib
import sys
import types
from contextlib import contextmanager

import numpy as np
import math

import torch
import torch.nn as nn

REF_SRC = "/root/reference/src/tacorl"


# --------------------------------------------------------------------------- shims
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class _ListConfig(list):
    pass


class _OmegaConf:
    @staticmethod
    def to_container(cfg, resolve=True):
        return copy.deepcopy(dict(cfg)) if isinstance(cfg, dict) else copy.deepcopy(cfg)

    @staticmethod
    def load(path):
        raise RuntimeError("OmegaConf.load is not available in the harness")


def _load_class(name):
    module_name, class_name = name.rsplit(".", 1)
    return getattr(importlib.import_module(module_name), class_name)


def _instantiate(cfg, *args, **kwargs):
    if not isinstance(cfg, dict) or "_target_" not in cfg:
        return cfg
    cfg = copy.deepcopy(dict(cfg))
    target = cfg.pop("_target_")
    cfg.pop("_recursive_", None)
    cfg.pop("_convert_", None)
    cfg.update(kwargs)
    return _load_class(target)(*args, **cfg)


class _LightningModule(nn.Module):
    """Just enough of pl.LightningModule for the reference modules to run a step."""

    def __init__(self, *a, **k):
        super().__init__()
        self.logged = {}
        self.current_epoch = 0
        self.automatic_optimization = True
        self._opts = None
        self.grad_log = []  # one {name: grad.clone()} per manual_backward call

    def log(self, name, value, **kw):
        self.logged[name] = float(value.detach()) if torch.is_tensor(value) else float(value)

    def save_hyperparameters(self, *a, **k):
        pass

    @property
    def device(self):
        return torch.device("cpu")

    def optimizers(self):
        if self._opts is None:
            o = self.configure_optimizers()
            self._opts = list(o) if isinstance(o, (list, tuple)) else [o]
        return self._opts

    def manual_backward(self, loss, **kw):
        loss.backward(**kw)
        self.grad_log.append(
            {n: p.grad.detach().clone() for n, p in self.named_parameters() if p.grad is not None}
        )


def install_shims():
    if "tacorl" in sys.modules and getattr(sys.modules["tacorl"], "_is_harness", False):
        return
    pkg = _mod("tacorl", _is_harness=True)
    pkg.__path__ = [REF_SRC]
    pkg.__file__ = REF_SRC + "/__init__.py"

    oc = _mod("omegaconf", DictConfig=_DictConfig, ListConfig=_ListConfig, OmegaConf=_OmegaConf)
    _mod("omegaconf.dictconfig", DictConfig=_DictConfig)
    _mod("omegaconf.omegaconf", DictConfig=_DictConfig, OmegaConf=_OmegaConf)
    _mod("omegaconf.listconfig", ListConfig=_ListConfig)
    del oc

    hy = _mod("hydra", main=lambda **kw: (lambda f: f))
    hy.utils = _mod("hydra.utils", instantiate=_instantiate)

    pl = _mod(
        "pytorch_lightning",
        LightningModule=_LightningModule,
        Callback=object,
        Trainer=object,
        LightningDataModule=object,
    )
    pl.loggers = _mod("pytorch_lightning.loggers", WandbLogger=object)
    pl.utilities = _mod("pytorch_lightning.utilities")
    pl.utilities.types = _mod("pytorch_lightning.utilities.types", STEP_OUTPUT=object)

    _mod("cv2")
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.transforms = _mod("torchvision.transforms")
    tv.transforms.functional = _mod(
        "torchvision.transforms.functional", adjust_contrast=None
    )
    gym = _mod("gym", Env=object)
    gym.envs = _mod("gym.envs")
    gym.envs.registration = _mod("gym.envs.registration", register=lambda **k: None)
    gym.spaces = _mod("gym.spaces")
    sb = _mod("stable_baselines3")
    sb.common = _mod("stable_baselines3.common")
    sb.common.utils = _mod("stable_baselines3.common.utils", set_random_seed=None)


# ----------------------------------------------------------------- noise recording
class NoiseTape:
    """Records every random draw of a step, in order, as (kind, tensor)."""

    def __init__(self):
        self.draws = []

    def add(self, kind, t):
        self.draws.append((kind, t.detach().clone()))

    def of_kind(self, kind):
        return [t for k, t in self.draws if k == kind]


@contextmanager
def record_noise(tape: NoiseTape):
    from torch.distributions import Normal

    o_sample, o_rsample = Normal.sample, Normal.rsample
    o_uniform, o_rand = torch.Tensor.uniform_, torch.rand

    def _eps(self, sample_shape):
        shape = self._extended_shape(sample_shape)
        eps = torch.randn(shape, dtype=self.loc.dtype)
        tape.add("normal", eps)
        return self.loc + eps * self.scale

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return _eps(self, sample_shape)

    def rsample(self, sample_shape=torch.Size()):
        return _eps(self, sample_shape)

    def uniform_(self, a=0.0, b=1.0, **kw):
        # record the U(0,1) base draw; apply the affine map the way torch does
        u = o_rand(self.shape, dtype=self.dtype)
        tape.add("uniform01", u)
        with torch.no_grad():
            self.copy_(u * (b - a) + a)
        return self

    def rand(*size, **kw):
        kw.pop("device", None)
        u = o_rand(*size, **kw)
        tape.add("rand", u)
        return u

    import torch.nn.functional as F

    o_dropout, o_sdpa = F.dropout, F.scaled_dot_product_attention

    def dropout(input, p=0.5, training=True, inplace=False):
        """F.dropout with the keep mask recorded (what nn.Dropout calls; torch's own kernel hides the mask).
        Same distribution and arithmetic as torch: keep ~ Bernoulli(1 - p), kept values scaled by 1 / (1 - p)."""
        if not training or p == 0.0:
            return input
        keep = torch.bernoulli(torch.full(input.shape, 1.0 - p))
        tape.add("dropout", keep.to(torch.uint8))
        return input * keep * (1.0 / (1.0 - p))

    def sdpa(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        """torch's scaled_dot_product_attention, written out so that its internal dropout on the attention
        probabilities goes through the recorded F.dropout above (nn.MultiheadAttention in train mode, need_weights=
        False, takes this route; its math path is softmax(q k^T * scale + mask) -> dropout -> @ v).
        check_sdpa_standin() pins it against torch's own implementation with dropout off."""
        assert attn_mask is None and not is_causal and not kw, "stand-in covers the plan-recognition use only"
        sc = (1.0 / math.sqrt(query.shape[-1])) if scale is None else scale
        att = torch.softmax((query * sc) @ key.transpose(-1, -2), dim=-1)
        att = dropout(att, dropout_p, True)
        return att @ value

    Normal.sample, Normal.rsample = sample, rsample
    torch.Tensor.uniform_, torch.rand = uniform_, rand
    F.dropout, F.scaled_dot_product_attention = dropout, sdpa
    try:
        yield tape
    finally:
        Normal.sample, Normal.rsample = o_sample, o_rsample
        torch.Tensor.uniform_, torch.rand = o_uniform, o_rand
        F.dropout, F.scaled_dot_product_attention = o_dropout, o_sdpa


def check_sdpa_standin():
    """The written-out attention used while recording equals torch's scaled_dot_product_attention (dropout off) and,
    through nn.MultiheadAttention in train mode, changes nothing: max abs difference of an encoder layer's output."""
    torch.manual_seed(0)
    layer = torch.nn.TransformerEncoderLayer(32, 8, dim_feedforward=64, dropout=0.0)
    layer.train()
    x = torch.randn(16, 3, 32)
    ref = layer(x)
    with record_noise(NoiseTape()):
        got = layer(x)
    return (got - ref).abs().max().item()


# ------------------------------------------------------------------ config dicts
P = "tacorl.networks."


def enc_cfg():
    return {
        "_target_": P + "visual_encoders.encoder.LMPVisionEncoder",
        "latent_dim": 32,
        "hidden_dim": 256,
        "normalize_output": False,
    }


def rep_cfg():
    return {
        "_target_": P + "representation.representation_network.LateFusion",
        "_recursive_": False,
        "networks": {"rgb_static": enc_cfg(), "rgb_gripper": enc_cfg()},
    }


def goal_cfg():
    return {
        "_target_": P + "visual_encoders.goal_encoder.VisualGoalEncoder",
        "in_features": None,
        "out_features": None,
        "hidden_size": 256,
        "activation_function": "ReLU",
        "last_layer_activation": "Identity",
    }


def actor_cfg(discrete_gripper=False):
    c = {
        "_target_": P + "actor_critic.actor.Actor",
        "_recursive_": False,
        "policy": {
            "_target_": P + "actor_critic.actor.MLPPolicy",
            "num_layers": 3,
            "hidden_dim": 256,
        },
    }
    if discrete_gripper:
        c["discrete_gripper"] = True
    return c


def critic_cfg():
    return {
        "_target_": P + "actor_critic.critic.Critic",
        "_recursive_": False,
        "q_network": {
            "_target_": P + "actor_critic.critic.MLPQNetwork",
            "num_layers": 3,
            "hidden_dim": 256,
            "last_layer_activation": "Identity",
        },
    }


def pr_cfg(latent_plan_dim, seq_len, dropout_p=0.0):
    return {
        "_target_": P + "plan_encoders.plan_recognition_transformer.PlanRecognitionTransformersNetwork",
        "num_heads": 8,
        "num_layers": 2,
        "encoder_hidden_size": 2048,
        "fc_hidden_size": 4096,
        "state_dim": None,
        "latent_plan_dim": latent_plan_dim,
        "min_std": 0.0001,
        "dropout_p": dropout_p,
        "encoder_normalize": False,
        "positional_normalize": False,
        "position_embedding": True,
        "max_position_embeddings": seq_len,
    }


def ad_cfg(latent_plan_dim, hidden_size=2048):
    return {
        "_target_": P + "action_decoders.action_decoder_logistic.ActionDecoderLogistic",
        "n_mixtures": 10,
        "num_layers": 2,
        "hidden_size": hidden_size,
        "out_features": 7,
        "act_max_bound": [1.0] * 7,
        "act_min_bound": [-1.0] * 7,
        "policy_rnn_dropout_p": 0.0,
        "num_classes": 10,
        "latent_plan_dim": latent_plan_dim,
        "rnn_model": "rnn_decoder",
        "include_goal": False,
    }


# --------------------------------------------------------------- module builders
def build_play_lmp(cams=("rgb_static",), latent_plan_dim=16, seq_len=16, ad_hidden=2048,
                   dropout_p=0.0, **kw):
    install_shims()
    from tacorl.modules.play_lmp.play_lmp_for_rl import PlayLMP

    cams = list(cams)
    return PlayLMP(
        plan_proposal=actor_cfg(),
        plan_recognition=pr_cfg(latent_plan_dim, seq_len, dropout_p),
        perceptual_encoder=rep_cfg(),
        goal_encoder=goal_cfg(),
        action_decoder=ad_cfg(latent_plan_dim, ad_hidden),
        plan_proposal_obs_modalities=cams,
        plan_proposal_goal_modalities=cams,
        plan_recognition_modalities=cams,
        action_decoder_modalities=cams,
        real_world=True,
        lr=1e-4,
        kl_beta=1e-3,
        **kw,
    )


TACORL_YAML = dict(  # config/module/tacorl.yaml:8-30
    action_decoder_lr=3e-4, actor_lr=1e-4, critic_lr=3e-4, discount=0.95,
    conservative_weight=1.0, reward_scale=10.0, n_action_samples=4, with_lagrange=True,
    deterministic_backup=True, bc_epochs=5, with_dr3=False, dr3_coefficient=0.03,
    with_vib=False, vib_coefficient=0.03,
)

CQL_YAML = dict(  # config/module/cql_offline_goal_cond.yaml:11-27
    discount=0.99, actor_lr=1e-4, critic_lr=3e-4, conservative_weight=1.0,
    n_action_samples=4, with_lagrange=True, reward_scale=10.0, deterministic_backup=False,
    bc_epochs=5, with_dr3=False, dr3_coefficient=0.03, with_vib=False, vib_coefficient=0.03,
)


def build_tacorl(play_lmp, finetune_action_decoder=False, **overrides):
    install_shims()
    import tacorl.modules.tacorl.tacorl as T

    T.load_pl_module_from_checkpoint = lambda *a, **k: play_lmp
    kw = dict(TACORL_YAML)
    kw.update(overrides)
    return T.TACORL(
        play_lmp_dir="/nonexistent",
        finetune_action_decoder=finetune_action_decoder,
        critic=critic_cfg(),
        critic_encoder=rep_cfg(),
        real_world=True,
        **kw,
    )


def build_cql(cams=("rgb_static",), action_dim=7, **overrides):
    install_shims()
    from tacorl.modules.cql.cql_offline_lightning import CQL_Offline

    kw = dict(CQL_YAML)
    kw.update(overrides)
    return CQL_Offline(
        actor=actor_cfg(discrete_gripper=True),
        critic=critic_cfg(),
        actor_encoder=rep_cfg(),
        critic_encoder=rep_cfg(),
        goal_encoder=goal_cfg(),
        real_world=True,
        obs_modalities=list(cams),
        goal_modalities=list(cams),
        action_dim=action_dim,
        **kw,
    )
