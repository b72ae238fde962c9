"""PlayLMP - drop-in for reference modules/play_lmp/play_lmp_for_rl.py:17-368 (seq-VAE:
perceptual encoder, plan recognition posterior, plan proposal prior, balanced KL,
logistic-mixture action decoder).  Select with
`module._target_=tacorl_amd.modules.play_lmp.play_lmp_for_rl.PlayLMP`.
"""
from pathlib import Path
from typing import List

import torch
import torch.nn as nn

from ... import _lib
from ..._lib import ACT_NONE, ACT_SILU
from ...engine import NetBlock
from ...init import init_views_
from ...networks.action_decoder import ActionDecoderLogistic
from ...networks.plan_recognition import PlanRecognition
from ..common import LoggerMixin, compute_flag, register_views, to_plain


class PlayLMP(LoggerMixin, nn.Module):
    def __init__(self, env={}, actor={}, plan_proposal={}, plan_recognition={}, perceptual_encoder={},
                 goal_encoder={}, action_decoder={}, transform_manager={}, dataloader={}, kl_beta: float = 1e-3,
                 kl_balancing: bool = True, add_random_plan_loss: bool = False, kl_alpha: float = 0.8,
                 lr: float = 1e-4, plan_proposal_obs_modalities: List[str] = [],
                 plan_proposal_goal_modalities: List[str] = [], plan_recognition_modalities: List[str] = [],
                 action_decoder_modalities: List[str] = [], real_world: bool = False, *args, device=None,
                 compute_dtype="f32", image_dtype="f32", **kwargs):
        super().__init__()
        _lib.lib()
        self.dev = torch.device(device if device is not None else "cuda:0")
        self.logged, self.current_epoch = {}, 0
        self.real_world, self.env, self.transform_manager = real_world, None, transform_manager
        self.add_random_plan_loss = add_random_plan_loss
        self.plan_proposal_obs_modalities = list(plan_proposal_obs_modalities)
        self.plan_proposal_goal_modalities = list(plan_proposal_goal_modalities)
        self.plan_recognition_modalities = list(plan_recognition_modalities)
        self.action_decoder_modalities = list(action_decoder_modalities)
        self.all_modalities = set(self.plan_proposal_obs_modalities + self.plan_proposal_goal_modalities
                                  + self.plan_recognition_modalities + self.action_decoder_modalities)
        if not self.all_modalities:
            raise ValueError("PlayLMP needs its *_modalities lists")
        self.lr, self.kl_beta, self.kl_balancing, self.kl_alpha = lr, kl_beta, kl_balancing, kl_alpha
        self.compute = compute_flag(compute_dtype)
        self.img_dtype = torch.bfloat16 if compute_flag(image_dtype) == _lib.BF16 else torch.float32
        self.pp_cfg, self.pr_cfg, self.ad_cfg = to_plain(plan_proposal), to_plain(plan_recognition), to_plain(action_decoder)
        self.build_networks()

    def build_networks(self):
        """reference play_lmp_for_rl.py:80-130."""
        cams = self.plan_proposal_obs_modalities
        if sorted(self.all_modalities) != sorted(set(cams)) or sorted(cams) != sorted(self.plan_proposal_goal_modalities):
            raise NotImplementedError("all PlayLMP modality lists must name the same cameras (in-scope configs)")
        pol = self.pp_cfg.get("policy", {})
        self.policy_layers, self.hidden = pol.get("num_layers", 2), pol.get("hidden_dim", 256)
        prc = {k: v for k, v in self.pr_cfg.items() if not k.startswith("_")}
        state_dim = 32 * len(self.plan_recognition_modalities)
        prc["state_dim"] = state_dim
        self.pr = PlanRecognition(device=self.dev, **prc)
        A = self.pr.latent_plan_dim
        E = 64 * len(cams)
        pdims = [E] + [self.hidden] * self.policy_layers + [2 * A]
        pn = [(f"actor.policy.fc_layers.{i}.weight", f"actor.policy.fc_layers.{i}.bias") for i in range(self.policy_layers)]
        parts = [("actor.policy.fc_mean", A), ("actor.policy.fc_log_std", A)]
        self.net = NetBlock(cams, cams, pdims, [ACT_SILU] * self.policy_layers + [ACT_NONE], pn, self.dev,
                            head_parts=parts, hidden=self.hidden)
        adc = {k: v for k, v in self.ad_cfg.items() if not k.startswith("_")}
        adc["state_dim"] = 32 * len(self.action_decoder_modalities)
        adc["goal_dim"] = 32 * len(cams)
        self.ad = ActionDecoderLogistic(device=self.dev, **adc)
        init_views_(self.net.views)
        init_views_(self.pr.blk.views)
        init_views_(self.ad.blk.views, rnn_hidden=self.ad.hidden)
        ren = {}
        for k, v in self.net.views.items():
            if k.startswith("encoder."):
                ren["perceptual_encoder." + k[len("encoder."):]] = v
            elif k.startswith("actor.policy."):
                ren["plan_proposal.policy." + k[len("actor.policy."):]] = v
            else:
                ren[k] = v
        register_views(self, "", ren)
        register_views(self, "plan_recognition.", self.pr.blk.views)
        register_views(self, "action_decoder.", self.ad.blk.views)
        for k, v in self.ad.buffers.items():
            self.action_decoder.register_buffer(k, v)

    @property
    def device(self):
        return self.dev

    def set_kl_beta(self, kl_beta):
        """reference :303-305."""
        self.kl_beta = kl_beta


def load_play_lmp(play_lmp_dir, epoch=-1, overwrite_cfg=None, device=None, compute_dtype="f32", image_dtype="f32"):
    """utils/networks.py:90-142 load_pl_module_from_checkpoint: first `*config.yaml` + `last.ckpt`
    (or `..._epoch_N...ckpt`) under `play_lmp_dir`."""
    import yaml

    d = Path(play_lmp_dir).expanduser()
    if d.is_file():
        ckpt, d = d, d.parent
    else:
        cks = list(d.rglob("*.ckpt"))
        if not cks:
            raise FileNotFoundError(f"no .ckpt under {d}")
        ckpt = next((c for c in cks if c.stem == "last"), None) if epoch == -1 else None
        if ckpt is None:
            ckpt = next((c for c in cks if f"epoch_{epoch}" in c.stem or f"epoch={epoch}" in c.stem), cks[-1])
    cfgs = list(d.rglob("*config.yaml"))
    if not cfgs:
        raise FileNotFoundError(f"no *config.yaml under {d}")
    cfg = yaml.safe_load(open(cfgs[0]))["module"]
    cfg = {k: v for k, v in cfg.items() if k not in ("_target_", "_recursive_")}
    cfg.update(overwrite_cfg or {})
    cfg.setdefault("real_world", True)
    mod = PlayLMP(device=device, compute_dtype=compute_dtype, image_dtype=image_dtype, **cfg)
    sd = torch.load(ckpt, map_location="cpu", weights_only=False)
    mod.load_state_dict(sd.get("state_dict", sd))
    return mod
