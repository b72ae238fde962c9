"""CQL_Offline - drop-in for reference modules/cql/cql_offline_lightning.py:24-574.

Same constructor kwargs (`_recursive_: False`: sub-configs arrive as dicts), same
`training_step / validation_step / configure_optimizers` surface, same `state_dict()` keys and
logged scalar names; the arithmetic runs in tacorl_amd.engine.ACEngine (HIP kernels).
Select it with `module._target_=tacorl_amd.modules.cql.cql_offline_lightning.CQL_Offline`.
"""
from typing import List

import torch
import torch.nn as nn

from ... import _lib
from ...engine import ACEngine
from ..common import GraphMixin, LoggerMixin, compute_flag, register_views, to_plain


class _OptimizerHandle:
    """What `configure_optimizers()` returns: the state lives in the engine's flat blocks."""

    def __init__(self, name, blk, lr):
        self.name, self.blk, self.lr = name, blk, lr

    def state_dict(self):
        return {"m": self.blk.m.clone(), "v": self.blk.v.clone(), "step": self.blk.step.clone(), "lr": self.lr}

    def load_state_dict(self, sd):
        self.blk.m.copy_(sd["m"]); self.blk.v.copy_(sd["v"]); self.blk.step.copy_(sd["step"])


class CQL_Offline(GraphMixin, LoggerMixin, nn.Module):
    def __init__(self, env={}, actor={}, critic={}, actor_encoder={}, critic_encoder={}, goal_encoder={},
                 transform_manager={}, discount: float = 0.99, tau: float = 0.005, actor_lr: float = 3e-4,
                 critic_lr: float = 3e-4, deterministic_backup: bool = False, reward_scale: float = 1.0,
                 bc_epochs: int = 0, clip_grad: bool = True, clip_grad_val: int = 1,
                 conservative_weight: float = 1.0, lagrange_thresh: float = 5.0, n_action_samples: int = 10,
                 temp: float = 1.0, with_lagrange: bool = False, with_dr3: bool = False,
                 dr3_coefficient: float = 0.03, with_vib: bool = False, vib_coefficient: float = 0.01,
                 real_world: bool = False, obs_modalities: List[str] = [], goal_modalities: List[str] = [],
                 action_dim: int = 7, *args, device=None, compute_dtype="f32", image_dtype="f32", world_size=1,
                 **kwargs):
        super().__init__()
        if with_dr3 or with_vib:
            raise NotImplementedError("DR3 / VIB regularisers are off in every in-scope config (SURVEY 8a A9)")
        _lib.lib()  # fail loudly, now, if the HIP extension is missing
        self.dev = torch.device(device if device is not None else "cuda:0")
        self.logged = {}
        self.current_epoch = 0
        self.automatic_optimization = False
        self.real_world = real_world
        self.env = None
        self.transform_manager = transform_manager
        self.bc_epochs, self.clip_grad = bc_epochs, clip_grad
        self.actor_lr, self.critic_lr = actor_lr, critic_lr
        self.with_lagrange = with_lagrange
        self.n_action_samples = n_action_samples
        self.compute = compute_flag(compute_dtype)
        self.img_dtype = torch.bfloat16 if compute_flag(image_dtype) == _lib.BF16 else torch.float32
        self.world_size = world_size
        self.log_every_n_steps = 1  # PL Trainer(log_every_n_steps=...) semantics: metrics are read back (one D2H
        self._step_count = 0        # sync) only on these steps
        self._graphs, self._use_graph = {}, False
        self._hp = dict(discount=discount, tau=tau, actor_lr=actor_lr, critic_lr=critic_lr,
                        deterministic_backup=deterministic_backup, reward_scale=reward_scale,
                        clip_grad_val=float(clip_grad_val) if clip_grad else 0.0,
                        conservative_weight=conservative_weight, lagrange_thresh=lagrange_thresh, temp=temp,
                        with_lagrange=with_lagrange, n=n_action_samples)
        self.actor_cfg, self.critic_cfg = to_plain(actor), to_plain(critic)
        if not real_world:
            # the reference builds the pybullet env here to read modalities / action space
            # (cql_offline_lightning.py:65-67,155-158); the simulator is out of scope, so the same
            # facts are read from the env config (config/env/goal_conditioned.yaml keys).
            e = to_plain(env) if env else {}
            obs_modalities = obs_modalities or list(e.get("modalities", []))
            goal_modalities = goal_modalities or list(e.get("goal_modalities", []))
        self.obs_modalities, self.goal_modalities = list(obs_modalities), list(goal_modalities)
        self.action_dim = action_dim
        # target entropy: -action_dim (real world, :93-94) or -prod(env.action_space.shape) = -7 (:96-98)
        self.target_entropy = -float(action_dim) if real_world else -7.0
        self.build_networks()

    # ------------------------------------------------------------------ construction
    def _arch(self):
        pol = self.actor_cfg.get("policy", {}) if self.actor_cfg else {}
        qn = self.critic_cfg.get("q_network", {}) if self.critic_cfg else {}
        for cfg, ok in ((pol, "MLPPolicy"), (qn, "MLPQNetwork")):
            t = cfg.get("_target_", ok)
            if not t.endswith(ok):
                raise NotImplementedError(f"{t}: only the default {ok} is on the hot path (SURVEY section 2 row 8)")
        return dict(policy_layers=pol.get("num_layers", 2), q_layers=qn.get("num_layers", 2),
                    hidden=pol.get("hidden_dim", 256),
                    discrete_gripper=bool(self.actor_cfg.get("discrete_gripper", False)))

    def build_networks(self):
        if not self.obs_modalities:
            raise ValueError("obs_modalities / goal_modalities are required (real_world=True style construction)")
        a = self._arch()
        if a["hidden"] != self.critic_cfg.get("q_network", {}).get("hidden_dim", 256):
            raise NotImplementedError("actor and critic hidden sizes must match")
        self._make_engine(self.obs_modalities, self.goal_modalities, self.action_dim, a)
        self._register()

    def _make_engine(self, cams, goal_cams, action_dim, a):
        hp = self._hp
        self.engine = ACEngine(
            cams, goal_cams, None, action_dim, None, self.dev, n=hp["n"], discount=hp["discount"], tau=hp["tau"],
            actor_lr=hp["actor_lr"], critic_lr=hp["critic_lr"], deterministic_backup=hp["deterministic_backup"],
            reward_scale=hp["reward_scale"], clip_grad_val=hp["clip_grad_val"],
            conservative_weight=hp["conservative_weight"], lagrange_thresh=hp["lagrange_thresh"], temp=hp["temp"],
            with_lagrange=hp["with_lagrange"], discrete_gripper=a["discrete_gripper"],
            target_entropy=self.target_entropy, policy_layers=a["policy_layers"], q_layers=a["q_layers"],
            hidden=a["hidden"], compute=self.compute, img_dtype=self.img_dtype, world_size=self.world_size)

    def _register(self):
        e = self.engine
        for name, blk in (("actor", e.actor), ("q1", e.q1), ("q2", e.q2), ("target_q1", e.tq1), ("target_q2", e.tq2)):
            register_views(self, name + ".", blk.views)
        self.log_alpha = nn.Parameter(e.log_alpha.param)
        if self.with_lagrange:
            self.log_alpha_prime = nn.Parameter(e.log_alpha_prime.param)
        self.actor.action_dim = self.action_dim
        self.actor.discrete_gripper = e.dg

    def sync_targets(self):
        """target.load_state_dict(q.state_dict()) (reference :226-227)."""
        e = self.engine
        e.tq1.param.copy_(e.q1.param); e.tq2.param.copy_(e.q2.param)

    @property
    def device(self):
        return self.dev

    def _stepped_blocks(self):
        """Parameter blocks the step's optimiser kernels write (see ops.touched / GraphMixin._run_segments)."""
        e = self.engine
        out = [e.actor.param, e.q1.param, e.q2.param, e.tq1.param, e.tq2.param, e.log_alpha.param, e.log_alpha_prime.param]
        ad = getattr(self, "ad", None)
        if ad is not None and getattr(self, "finetune_action_decoder", False):
            out.append(ad.blk.param)
        return out

    def named_gradients(self):
        e = self.engine
        out = {}
        for name, blk in (("actor", e.actor), ("q1", e.q1), ("q2", e.q2)):
            out.update({f"{name}.{k}": v for k, v in blk.grad_views.items()})
        out["log_alpha"] = e.log_alpha.grad
        out["log_alpha_prime"] = e.log_alpha_prime.grad
        return out

    # ---------------------------------------------------------------------- stepping
    def overwrite_batch(self, batch):
        """reference :118-147 (shape/dtype normalisation of the transition batch)."""
        obs, nxt = batch["observations"], batch["next_observations"]
        return obs, batch["actions"].float(), nxt, batch["rewards"].float(), batch["terminals"].int()

    def _stage(self, obs, goal, nxt, action, reward, done, noise, nchw=True):
        e = self.engine
        B = action.shape[0]
        hw = {}
        if obs[e.cams[0]].dtype == torch.uint8:  # the dataset's uint8 HWC frames: normalised on the GPU (engine.load_images)
            nchw = False
        for c in e.cams:
            t = obs[c]
            hw[c] = tuple(t.shape[-2:]) if nchw else tuple(t.shape[-3:-1])
        e.ensure_batch(B, hw)
        for c in e.cams:
            e.load_images(c, obs[c].to(self.dev), goal[c].to(self.dev), nxt[c].to(self.dev), nchw=nchw)
        e.load_transition(action.to(self.dev), reward.to(self.dev), done.to(self.dev))
        e.set_noise(noise)

    def compute_update(self, batch, optimize: bool = True, log_type: str = "train", noise=None):
        obs, action, nxt, reward, done = batch
        self._stage(obs["observation"], obs["goal"], nxt["observation"], action, reward, done, noise)
        bc = self.current_epoch < self.bc_epochs
        e = self.engine
        self._run_segments(("cql", e.B, bc, optimize),
                           [lambda: e.phase_a(), lambda: e.phase_b(bc, optimize), lambda: e.phase_c(optimize)],
                           [e.allreduce_alpha, e.allreduce_grads])
        self._publish_logs(log_type)

    def _publish_logs(self, log_type, extra=()):
        self._step_count += 1
        if self.log_every_n_steps > 1 and self._step_count % self.log_every_n_steps:
            return
        m = self.engine.metrics()
        skip = {"action_loss"} | (set() if self.with_lagrange else {"alpha_prime", "alpha_prime_loss"})
        for k, v in m.items():
            if k not in skip or k in extra:
                self.log(f"{log_type}/{k}", v, on_step=True)

    def training_step(self, batch, batch_idx=0, noise=None):
        self.compute_update(self.overwrite_batch(batch), optimize=True, log_type="train", noise=noise)

    def validation_step(self, batch, *args, noise=None, **kwargs):
        self.compute_update(self.overwrite_batch(batch), optimize=False, log_type="validation", noise=noise)

    def configure_optimizers(self):
        """Order [alpha, actor, q1, q2, (alpha')] as the reference (:553-574)."""
        e = self.engine
        o = [_OptimizerHandle("alpha", e.log_alpha, self.actor_lr), _OptimizerHandle("actor", e.actor, self.actor_lr),
             _OptimizerHandle("q1", e.q1, self.critic_lr), _OptimizerHandle("q2", e.q2, self.critic_lr)]
        if self.with_lagrange:
            o.append(_OptimizerHandle("alpha_prime", e.log_alpha_prime, self.critic_lr))
        return o
