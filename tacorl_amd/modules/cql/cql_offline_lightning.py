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
from ... import dist as _D
from ...engine import ACEngine
from ...lightning import LightningModuleBase
from ..common import GraphMixin, ModuleMixin, broadcast_blocks, register_views, to_plain


class CQL_Offline(GraphMixin, ModuleMixin, LightningModuleBase):
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
        self._init_runtime(device, compute_dtype, image_dtype, world_size)
        self.automatic_optimization = False  # reference :115
        self.save_hyperparameters(ignore=["play_lmp"])  # reference :116
        self.real_world = real_world
        self.env = None
        self.transform_manager = transform_manager
        self.bc_epochs, self.clip_grad = bc_epochs, clip_grad
        self.actor_lr, self.critic_lr = actor_lr, critic_lr
        self.with_lagrange = with_lagrange
        self.n_action_samples = n_action_samples
        self._hp = dict(discount=discount, tau=tau, actor_lr=actor_lr, critic_lr=critic_lr,
                        deterministic_backup=deterministic_backup, reward_scale=reward_scale,
                        clip_grad_val=float(clip_grad_val) if clip_grad else 0.0,
                        conservative_weight=conservative_weight, lagrange_thresh=lagrange_thresh, temp=temp,
                        with_lagrange=with_lagrange, n=n_action_samples)
        self.actor_cfg, self.critic_cfg = to_plain(actor), to_plain(critic)
        self.actor_encoder_cfg, self.critic_encoder_cfg = to_plain(actor_encoder), to_plain(critic_encoder)
        self.goal_encoder_cfg = to_plain(goal_encoder)
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
        if _D.collectives_on(world_size):
            self.sync_from_rank0()

    # ------------------------------------------------------------------ construction
    def _arch(self):
        from .. import cfgcheck

        pol = cfgcheck.check_actor(self.actor_cfg, "actor")
        qn = cfgcheck.check_critic(self.critic_cfg, "critic")
        return dict(policy_layers=pol.get("num_layers", 2), q_layers=qn.get("num_layers", 2),
                    hidden=pol.get("hidden_dim", 256),
                    discrete_gripper=bool(self.actor_cfg.get("discrete_gripper", False)))

    def build_networks(self):
        if not self.obs_modalities:
            raise ValueError("obs_modalities / goal_modalities are required (real_world=True style construction)")
        from .. import cfgcheck

        a = self._arch()
        if a["hidden"] != (self.critic_cfg or {}).get("q_network", {}).get("hidden_dim", 256):
            raise NotImplementedError("actor and critic hidden sizes must match")
        cfgcheck.check_representation(self.actor_encoder_cfg, "actor_encoder", self.obs_modalities)
        cfgcheck.check_representation(self.critic_encoder_cfg, "critic_encoder", self.obs_modalities)
        cfgcheck.check_goal_encoder(self.goal_encoder_cfg, "goal_encoder", a["hidden"])
        self._make_engine(self.obs_modalities, self.goal_modalities, self.action_dim, a)
        # fresh networks start from the reference modules' torch initialisers (tacorl_amd/init.py); targets are
        # copies of the critics (reference :226-227)
        from ...init import init_views_

        e = self.engine
        for blk in (e.actor, e.q1, e.q2):
            init_views_(blk.views)
        self.sync_targets()
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
        self._pv = {}  # block name -> {view name: Parameter}: what the BlockAdams are built from
        for name, blk in (("actor", e.actor), ("q1", e.q1), ("q2", e.q2), ("target_q1", e.tq1), ("target_q2", e.tq2)):
            self._pv[name] = register_views(self, name + ".", blk.views)
        self.log_alpha = nn.Parameter(e.log_alpha.param)
        if self.with_lagrange:
            self.log_alpha_prime = nn.Parameter(e.log_alpha_prime.param)
        from ..inference import attach_rollout_surface

        attach_rollout_surface(self, e.actor, e.cams, self.goal_modalities, self.action_dim, e.dg)

    def _all_blocks(self):
        e = self.engine
        out = [e.actor, e.q1, e.q2, e.tq1, e.tq2, e.log_alpha, e.log_alpha_prime]
        for extra in ("lmp_net",):
            if getattr(self, extra, None) is not None:
                out.append(getattr(self, extra))
        for extra in ("pr", "ad"):
            if getattr(self, extra, None) is not None:
                out.append(getattr(self, extra).blk)
        return out

    def sync_from_rank0(self):
        """Broadcast rank 0's parameters and optimiser state (PL's DDP wrap does this for the reference module)."""
        broadcast_blocks(self._all_blocks())
        from ... import ops
        ops.touched(*[b.param for b in self._all_blocks()])

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

    def _derived_stale(self):
        """GraphMixin asks before every replay: are the engine's Adam-written bf16 mirrors still those of the parameters?"""
        e = self.engine
        return (e.adam_writes_mirrors and e.mirrors_stale()) or e.packs_stale()

    def _after_replay_touch(self):
        """A replay has just run (its optimiser launch rewrote the mirrors, its tail re-packed the encoders' conv weights)
        and the version counters were bumped for it."""
        self.engine.mirrors_written()
        self.engine.packs_written()

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
        self._sync_lrs()
        self._stage(obs["observation"], obs["goal"], nxt["observation"], action, reward, done, noise)
        bc = self.current_epoch < self.bc_epochs
        e = self.engine
        self._run_segments(("cql", e.B, bc, optimize),
                           [lambda: e.phase_a(optimize=optimize), lambda: e.phase_b(bc, optimize), lambda: e.phase_c(optimize)],
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
        self._tick_optimizers()

    def validation_step(self, batch, *args, noise=None, **kwargs):
        self.compute_update(self.overwrite_batch(batch), optimize=False, log_type="validation", noise=noise)

    def _net_adam(self, name, blk, key, lr):
        return self._make_adam(name, [(blk, self._pv[key], blk.views_of(blk.m), blk.views_of(blk.v))], lr)

    def _scalar_adam(self, name, sc, param, lr):
        return self._make_adam(name, [(sc, {"v": param}, {"v": sc.m}, {"v": sc.v})], lr)

    def configure_optimizers(self):
        """Order [alpha, actor, q1, q2, (alpha')] as the reference (:553-574).  torch.optim.Optimizer subclasses whose
        state is the engine's Adam blocks; the update itself runs inside training_step (manual optimisation)."""
        e = self.engine
        o = [self._scalar_adam("alpha", e.log_alpha, self.log_alpha, self.actor_lr),
             self._net_adam("actor", e.actor, "actor", self.actor_lr),
             self._net_adam("q1", e.q1, "q1", self.critic_lr), self._net_adam("q2", e.q2, "q2", self.critic_lr)]
        if self.with_lagrange:
            o.append(self._scalar_adam("alpha_prime", e.log_alpha_prime, self.log_alpha_prime, self.critic_lr))
        self._optimizers = o
        return o

    def _sync_lrs(self):
        """Learning rates edited on the optimisers' param_groups (schedulers, hand edits) reach the kernels: they are
        launch arguments, so a change also drops the captured graphs."""
        opts = getattr(self, "_optimizers", None)
        if not opts:
            return
        hp, by = self.engine.hp, {o.name: o.lr for o in opts}
        new = dict(actor_lr=by["actor"], critic_lr=by["q1"])
        if by["alpha"] != by["actor"] or by["q2"] != by["q1"] or by.get("alpha_prime", by["q1"]) != by["q1"]:
            raise NotImplementedError("alpha/actor share actor_lr and q1/q2/alpha' share critic_lr (reference :553-574)")
        if "action_decoder" in by and by["action_decoder"] != self.action_decoder_lr:
            self.action_decoder_lr, self._graphs = by["action_decoder"], {}
        if any(hp[k] != v for k, v in new.items()):
            hp.update(new)
            self.actor_lr, self.critic_lr, self._graphs = new["actor_lr"], new["critic_lr"], {}
