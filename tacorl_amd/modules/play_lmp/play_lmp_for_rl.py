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
from ... import dist as _D
from ..._lib import ACT_NONE, ACT_SILU
from ...engine import NetBlock
from ...init import init_views_
from ...networks.action_decoder import ActionDecoderLogistic
from ...networks.plan_recognition import PlanRecognition
from ...lightning import LightningModuleBase
from .. import cfgcheck
from ..common import GraphMixin, ModuleMixin, register_views, to_plain


class PlayLMP(GraphMixin, ModuleMixin, LightningModuleBase):
    def __init__(self, env={}, actor={}, plan_proposal={}, plan_recognition={}, perceptual_encoder={},
                 goal_encoder={}, action_decoder={}, transform_manager={}, dataloader={}, kl_beta: float = 1e-3,
                 kl_balancing: bool = True, add_random_plan_loss: bool = False, kl_alpha: float = 0.8,
                 lr: float = 1e-4, plan_proposal_obs_modalities: List[str] = [],
                 plan_proposal_goal_modalities: List[str] = [], plan_recognition_modalities: List[str] = [],
                 action_decoder_modalities: List[str] = [], real_world: bool = False, *args, device=None,
                 compute_dtype="f32", image_dtype="f32", world_size=1, **kwargs):
        super().__init__()
        self._init_runtime(device, compute_dtype, image_dtype, world_size)
        # The reference leaves optimisation to PL (one Adam over everything, :362-368).  Here forward, backward and
        # the fused Adam are one kernel sequence inside training_step, i.e. manual optimisation in PL's terms.
        self.automatic_optimization = False
        self.save_hyperparameters()  # reference :79
        self.real_world, self.env, self.transform_manager = real_world, None, transform_manager
        self.dataloader = dataloader
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
        self.pp_cfg, self.pr_cfg, self.ad_cfg = to_plain(plan_proposal), to_plain(plan_recognition), to_plain(action_decoder)
        self.pe_cfg, self.ge_cfg = to_plain(perceptual_encoder), to_plain(goal_encoder)
        self.build_networks()
        if _D.collectives_on(self.world_size):
            self.sync_from_rank0()

    def build_networks(self):
        """reference play_lmp_for_rl.py:80-130."""
        cams = self.plan_proposal_obs_modalities
        if sorted(self.all_modalities) != sorted(set(cams)) or sorted(cams) != sorted(self.plan_proposal_goal_modalities):
            raise NotImplementedError("all PlayLMP modality lists must name the same cameras (in-scope configs)")
        pol = cfgcheck.check_actor(self.pp_cfg, "plan_proposal")
        self.policy_layers, self.hidden = pol.get("num_layers", 2), pol.get("hidden_dim", 256)
        cfgcheck.check_representation(self.pe_cfg, "perceptual_encoder", cams)
        cfgcheck.check_goal_encoder(self.ge_cfg, "goal_encoder", self.hidden)
        prc = cfgcheck.check_plan_recognition(self.pr_cfg, "plan_recognition")
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
        adc = cfgcheck.check_action_decoder(self.ad_cfg, "action_decoder")
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
        pv = register_views(self, "", ren)
        back = {v_: k_ for k_, v_ in zip(self.net.views, ren)}  # renamed -> block view name
        self._pv = {"net": {back[k]: p for k, p in pv.items()},
                    "pr": register_views(self, "plan_recognition.", self.pr.blk.views),
                    "ad": register_views(self, "action_decoder.", self.ad.blk.views)}
        for k, v in self.ad.buffers.items():
            self.action_decoder.register_buffer(k, v)
        # rollout surface (evaluation/rollout_manager.py PlayLMP manager: plan_proposal.get_actions, action_decoder.act)
        from ..inference import ActorSurface, EncoderRunner, state_from_observation

        surf = ActorSurface(self, self.net, cams, cams, A, False)
        self.__dict__["_actor_surface"] = surf
        self.plan_proposal.get_actions = surf.get_actions
        self.plan_proposal.action_dim = A
        runner = EncoderRunner(self)
        self.__dict__["_pe_runner"] = runner
        self.perceptual_encoder.get_state_from_observation = (
            lambda observation, modalities=None: state_from_observation(self, runner, self.net, observation,
                                                                        list(modalities or cams)))
        self.action_decoder.clear_hidden_state = self.ad.clear_hidden_state
        self.action_decoder.act = lambda latent_plan, perceptual_emb, latent_goal=None, noise=None: self.ad.act(
            latent_plan, perceptual_emb, latent_goal, noise=noise, compute=self.compute)

    def sync_from_rank0(self):
        """Broadcast rank 0's parameters and optimiser state (PL's DDP wrap does this for the reference module)."""
        from ... import ops
        from ..common import broadcast_blocks

        blocks = [self.net, self.pr.blk, self.ad.blk]
        broadcast_blocks(blocks)
        ops.touched(*[b.param for b in blocks])

    def set_kl_beta(self, kl_beta):
        """reference :303-305."""
        self.kl_beta = kl_beta


def load_play_lmp(play_lmp_dir, epoch=-1, overwrite_cfg=None, device=None, compute_dtype="f32", image_dtype="f32",
                  unsafe_pickle=False):
    """utils/networks.py:90-117 load_pl_module_from_checkpoint: the run directory's first `*config.yaml` (loaded as
    a whole and resolved, so `${latent_plan_dim}`-style interpolations arrive as values) + `last.ckpt` or the
    checkpoint of epoch N.  The checkpoint is read with weights_only=True (a state_dict needs nothing else);
    unsafe_pickle=True opts into a full unpickle for checkpoints that carry arbitrary objects."""
    from ..common import find_checkpoint, load_resolved_yaml

    d = Path(play_lmp_dir).expanduser()
    if d.is_file():
        if d.suffix != ".ckpt":
            raise ValueError("File must have .ckpt extension")
        ckpt, d = d, d.parent
    elif d.is_dir():
        ckpt = find_checkpoint(d, epoch)
    else:
        raise ValueError(f"not valid file path: {d}")
    cfgs = list(d.rglob("*config.yaml"))
    if not cfgs and ckpt.parent == d:  # a checkpoint file was named: the run's .hydra/ sits beside its model_ckpts/
        cfgs = list(d.parent.rglob("*config.yaml"))
    if not cfgs:
        raise FileNotFoundError(f"no *config.yaml under {d}")
    cfg = load_resolved_yaml(cfgs[0])["module"]
    cfg = {k: v for k, v in cfg.items() if k not in ("_target_", "_recursive_")}
    cfg.update(overwrite_cfg or {})
    cfg.setdefault("real_world", True)
    mod = PlayLMP(device=device, compute_dtype=compute_dtype, image_dtype=image_dtype, **cfg)
    import os

    unsafe_pickle = unsafe_pickle or os.environ.get("TACORL_UNSAFE_PICKLE", "") not in ("", "0")
    try:
        sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    except Exception as e:
        # real pytorch_lightning checkpoints carry hyper_parameters (an OmegaConf DictConfig) and callback state, which
        # the weights-only unpickler rejects; loading them executes whatever the file's pickle stream says
        if not unsafe_pickle:
            raise RuntimeError(
                f"{ckpt}: not loadable with torch.load(weights_only=True) ({type(e).__name__}: {e}).  If you trust the file, "
                "pass TACORL(..., lmp_unsafe_pickle=True) / load_play_lmp(..., unsafe_pickle=True) or set "
                "TACORL_UNSAFE_PICKLE=1 to unpickle it in full.") from e
        sd = torch.load(ckpt, map_location="cpu", weights_only=False)
    mod.load_state_dict(sd.get("state_dict", sd))
    return mod


# --------------------------------------------------------------------------- training step
def _playlmp_ensure(self, B, T, hw):
    key = (B, T, tuple(sorted(hw.items())))
    if getattr(self, "_shape", None) == key:
        return
    from ... import ops

    ops.note_alloc()
    dev, cams, net = self.dev, self.plan_proposal_obs_modalities, self.net
    f = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    R, Ec = B * T, 32 * len(cams)
    self.frames = {c: torch.zeros(R, *hw[c], 3, device=dev, dtype=self.img_dtype) for c in cams}
    self.f_out = {c: f(R, 32) for c in cams}
    self.f_act = {c: f(ops.encoder_act_layout(R, *hw[c])[1]) for c in cams}
    self.f_dout = {c: f(R, 32) for c in cams}
    self.emb, self.d_emb = f(R, Ec), f(R, Ec)
    self.gin, self.dgin = f(B, Ec), f(B, Ec)
    self.gact = f(ops.mlp_act_layout(B, net.genc_dims, net.genc_acts)[2])
    self.g_yoff = ops.mlp_act_layout(B, net.genc_dims, net.genc_acts)[1][-1]
    self.S, self.dS = f(B, 2 * Ec), f(B, 2 * Ec)
    self.pact = f(ops.mlp_act_layout(B, net.head_dims, net.head_acts)[2])
    self.p_yoff = ops.mlp_act_layout(B, net.head_dims, net.head_acts)[1][-1]
    A = self.pr.A
    self.d_head_pp, self.d_head_pr = f(B, 2 * A), f(B, 2 * A)
    self.plan, self.rplan, self.d_plan = f(B, A), f(B, A), f(B, A)
    self.noise = dict(eps_plan=f(B, A), u_plan=f(B, A))
    self.logs = f(16)  # 0 kl, 1 kl_scaled, 2 action_loss, 3 gripper_acc, 4 rand action_loss, 5 rand gripper_acc
    self._shape = key


def _playlmp_step(self, batch, noise=None, optimize=True, log_type="train", nchw=True):
    """PlayLMP.training_step + the single Adam (reference play_lmp_for_rl.py:200-257,307-317,362-368)."""
    from ... import ops
    from ..._lib import BF16, F32, call, ptr
    from ...data.replay import wait_ready

    wait_ready(batch)  # a replay batch's small tables travel on the feeder's copy stream
    rp = batch.get("replay")  # frames by index out of a uint8 dataset (data/replay.py HbmReplay.batch(fused=True))
    states = rp["frames"] if rp is not None else batch["states"]
    cams, net, pr, ad = self.plan_proposal_obs_modalities, self.net, self.pr, self.ad
    opts = getattr(self, "_optimizers", None)
    if opts and opts[0].lr != self.lr:  # lr edited on the optimiser (scheduler): a launch argument -> new captures
        self.lr, self._graphs = opts[0].lr, {}
    if rp is not None:
        B, T, u8, nchw = rp["B"], rp["T"], True, False
        hw = {c: tuple(v.shape[1:3]) for c, v in states.items()}
    else:
        B, T = next(iter(states.values())).shape[:2]
        u8 = next(iter(states.values())).dtype == torch.uint8  # the dataset's uint8 HWC frames: normalised by the pack
        if u8:
            nchw = False
        hw = {c: (tuple(v.shape[-2:]) if nchw else tuple(v.shape[-3:-1])) for c, v in states.items()}
    src_hw = dict(hw)  # the frames as stored; an augmentation spec with a Resize stage sets the encoders' geometry
    rs = (batch.get("aug") or {}).get("resize") or {}
    if rs:
        if not u8:
            raise ValueError("aug['resize'] needs the dataset's uint8 frames (the resize is part of the uint8 pack)")
        hw = {c: tuple(rs.get(c, hw[c])) for c in hw}
    _playlmp_ensure(self, B, T, hw)
    R, Ec, A, cd = B * T, 32 * len(cams), pr.A, self.compute
    xd = BF16 if self.img_dtype == torch.bfloat16 else F32
    gs = 1.0 / getattr(self, "world_size", 1)
    for k, buf in self.noise.items():
        if noise is not None:
            buf.copy_(noise[k].reshape(buf.shape))
        elif k.startswith("eps"):
            buf.normal_()
        else:
            buf.uniform_()
    # plan-recognition dropout is live in train mode (reference transformer.yaml:9, nn.Module.training): its keep masks
    # are drawn (or injected: noise["dropout"]) here, with the step's other noise
    self._pr_train = bool(self.training and pr.dropout_p > 0)
    if self._pr_train:
        pr.stage_dropout(B, T, noise.get("dropout") if noise is not None else None)
    # ---- eager staging: frames into the fixed NHWC buffers, actions into a fixed buffer
    for c in cams:
        H, W = hw[c]
        v = states[c].to(self.dev)
        aug = batch.get("aug") if u8 else None
        Hs, Ws = src_hw[c]
        if rp is not None:  # window frames by index straight out of the dataset: gather + pack in one pass
            ids = rp["ids"]
            job = (v.data_ptr(), 3 * Hs * Ws, self.frames[c].data_ptr(), R, ids.data_ptr(), 1)
            if aug is None:
                ops.pack_images_u8_gather_batch([job], xd, H, W)
            else:
                st = aug["states"][c]
                flat = lambda t: None if t is None else t.reshape(R, t.shape[-1]).contiguous()  # noqa: E731
                ops.pack_images_u8_resize_aug_batch([job + (flat(st.get("shift")), flat(st.get("jitter")))], xd, (Hs, Ws), H, W,
                                                    aug["pad"][c])
        elif aug is not None:  # train-time augmentations on the way in, draws as device tables
            st = aug["states"][c]
            flat = lambda t: None if t is None else t.reshape(R, t.shape[-1]).contiguous()  # noqa: E731
            ops.pack_images_u8_resize_aug_batch([(v.data_ptr(), 3 * Hs * Ws, self.frames[c].data_ptr(), R, None, 1,
                                                  flat(st.get("shift")), flat(st.get("jitter")))], xd, (Hs, Ws), H, W,
                                                aug["pad"][c])
        elif u8:
            if (H * W * 3) % 16 or v.data_ptr() % 16 or not v.is_contiguous():
                raise ValueError("uint8 frames: contiguous, 16-byte aligned, H*W*3 a multiple of 16")
            ops.pack_images_u8_batch([(v.data_ptr(), 3 * H * W, self.frames[c].data_ptr(), R)], xd, H, W)
        elif nchw and (H * W) % 4 == 0 and v.data_ptr() % 16 == 0:
            ops.pack_images_batch([(v.data_ptr(), 3 * H * W, self.frames[c].data_ptr(), R)], xd, H, W)
        else:
            call("tacorl_pack_images", ptr(v), 3 * H * W, int(nchw), ptr(self.frames[c]), xd, R, 3, H, W, ops.stream())
    if getattr(self, "_acts", None) is None or self._acts.shape != batch["actions"].shape:
        ops.note_alloc()
        self._acts = torch.zeros(*batch["actions"].shape, device=self.dev)
    self._acts.copy_(batch["actions"])
    acts = self._acts

    from ... import dist as D

    reduce_on = optimize and D.collectives_on(getattr(self, "world_size", 1))
    if reduce_on:
        _grad_arena(self)  # (before anything is captured: the backward kernels write into the arena's slices)

    def fwd_bwd():
        _playlmp_fwd_bwd(self, B, T, hw, acts, gs)

    def opt():
        if optimize:
            ops.adam_step_batch([(blk.param, blk.grad, blk.m, blk.v, self.lr, 0.0, blk.step, None, 0.0)
                                 for blk in (net, pr.blk, ad.blk)])
        ops.mark("end")

    def reduce_grads():
        if reduce_on:
            D.all_reduce_sum_(_grad_arena(self))  # ONE collective: [encoders + proposal | plan recognition | action decoder]

    # (a graph replay runs no python: announce the optimiser's writes to torch's version counters - ops.touched)
    self._stepped_blocks = (lambda: [net.param, pr.blk.param, ad.blk.param]) if optimize else None
    self._run_segments(("playlmp", B, T, tuple(sorted(hw.items())), optimize, self._pr_train), [fwd_bwd, opt], [reduce_grads])
    # metrics cross to the host (a device synchronisation) only on logging steps - Trainer(log_every_n_steps), as the
    # other two modules do; in between the step returns the last total it read
    self._step_count += 1
    # Deviation from the reference (which logs and returns the total on every step, :307-317): with log_every_n_steps = k > 1
    # the train/* values - and their on_epoch means - come from every k-th step, and the steps in between return the last
    # total that was read (never None: the first step always reads).
    if (optimize and self.log_every_n_steps > 1 and self._step_count % self.log_every_n_steps
            and self.__dict__.get("_last_total") is not None):
        return self.__dict__["_last_total"]
    # (sync_dist=True in the reference, :162,183,292-339: the per-rank batch means are averaged over the ranks first)
    logs, div = D.reduce_logs_(self.logs, getattr(self, "world_size", 1))
    lg = [x / div for x in logs.cpu().tolist()]
    names = ["kl_loss", "kl_loss_scaled", "action_loss", "gripper_accuracy", "random_plan_action_loss",
             "random_plan_gripper_accuracy"]
    for k, v in zip(names, lg):
        self.log(f"{log_type}/{k}", v, on_step=True, on_epoch=True, sync_dist=True)
    total = lg[1] + lg[2]
    self.log(f"{log_type}/total_loss", total, on_step=True, on_epoch=True, sync_dist=True)
    self.__dict__["_last_total"] = total
    return total



def _playlmp_pp_forward(self, B, T, Ec, A, cd):
    """Plan proposal: goal encoder on emb[:, -1], policy head on [emb[:, 0] | goal]; returns the (B, 2A) head view."""
    from ... import ops
    from ..._lib import BF16, call

    net = self.net
    # round 5: both MLP inputs - emb[:, -1] for the goal encoder, [emb[:, 0] | goal_enc] for the policy head - are gathered by
    # the fused forwards where their producers left them (ops.mlp_fwd_gather, as the actor-critic engine does): three copy
    # launches off this chain; the assembled rows the weight gradients read are written from the forward's registers
    gather = (cd == BF16 and getattr(self, "fused_mlps", True) and getattr(self, "pp_gather", True)
              and ops.mlp_fwd_gather_ok([B], net.genc_dims, net.genc_acts, Ec, cd, False)
              and ops.mlp_fwd_gather_ok([B], net.head_dims, net.head_acts, 2 * Ec, cd, False))
    if not gather:
        ops.copy_cols(self.emb, (T - 1) * Ec, T * Ec, self.gin, 0, Ec, B, Ec)  # pp_goal input = emb[:, -1]
    # bf16 mode: the goal encoder and the plan proposal's policy head run as the single-launch MLP kernels (forward, input
    # gradients, weight gradients) on a bf16 mirror of their weights, as in the actor-critic engine.  (Per layer they are 7
    # generic GEMM launches on this chain: hidden behind the random-plan decoder pass while that was a pass of its own -
    # no gain then -, on the critical path since it rides in the real pass.)
    pb_g = pb_h = None
    if cd == BF16 and getattr(self, "fused_mlps", True):
        import ctypes as C
        call("tacorl_to_bf16_batch", 1, ops.ptr_array([net.genc()]), ops.ptr_array([net.genc_bf16()]),
             (C.c_long * 1)(net.size - net.genc_off), ops.stream())
        pb_g, pb_h = [net.genc_bf16()], [net.head_bf16()]
    self._pp_bf16 = pb_g
    if gather:
        ops.mlp_fwd_gather([[(self.emb, (T - 1) * Ec, T * Ec, 0, 0)]], [self.gin], Ec, [net.genc()], pb_g, [self.gact], [B],
                           net.genc_dims, net.genc_acts)
        ops.mlp_fwd_gather([[(self.emb, 0, T * Ec, 0, 0), (self.gact, self.g_yoff, Ec, Ec, 0)]], [self.S], 2 * Ec, [net.head()], pb_h,
                           [self.pact], [B], net.head_dims, net.head_acts)
        return self.pact[self.p_yoff: self.p_yoff + B * 2 * A]
    ops.mlp_fwd([self.gin], Ec, [net.genc()], [self.gact], [B], net.genc_dims, net.genc_acts, cd, params_bf16=pb_g)
    ops.copy_cols(self.emb, 0, T * Ec, self.S, 0, 2 * Ec, B, Ec)  # pp_state = emb[:, 0]
    ops.copy_cols(self.gact, self.g_yoff, Ec, self.S, Ec, 2 * Ec, B, Ec)
    ops.mlp_fwd([self.S], 2 * Ec, [net.head()], [self.pact], [B], net.head_dims, net.head_acts, cd, params_bf16=pb_h)
    return self.pact[self.p_yoff: self.p_yoff + B * 2 * A]


def _playlmp_fwd_bwd(self, B, T, hw, acts, gs):
    """Device side of the PlayLMP step up to the gradients (fixed buffers only: hipGraph-capturable)."""
    from ... import ops
    from ..._lib import BF16, F32, call, ptr

    cams, net, pr, ad = self.plan_proposal_obs_modalities, self.net, self.pr, self.ad
    R, Ec, A, cd = B * T, 32 * len(cams), pr.A, self.compute
    xd = BF16 if self.img_dtype == torch.bfloat16 else F32
    ops.mark("start")
    for j, c in enumerate(cams):
        H, W = hw[c]
        fused = (cd == BF16 and xd == BF16 and bool(ops.L.lib().tacorl_encoder_fused_supported(H, W))
                 and ops.L.lib().tacorl_encoder_bwd_fused_ws_bytes(1, ops.int_array([R]), H, W) > 0)  # forward and backward
        if fused:  # one launch for the whole encoder, activations saved for the backward (encoder_fused.hip)
            if getattr(self, "_wpk", None) is None:
                self._wpk = {}
            if c not in self._wpk:
                ops.note_alloc()
                self._wpk[c] = torch.empty(ops.L.lib().tacorl_encoder_fused_wpk_bytes(), dtype=torch.uint8, device=self.dev)
            call("tacorl_encoder_pack_weights", 1, ops.ptr_array([net.enc(c)]), ops.ptr_array([self._wpk[c]]), ops.stream())
            call("tacorl_encoder_fwd_fused", 1, ops.ptr_array([self.frames[c]]), ops.ptr_array([self._wpk[c]]),
                 ops.ptr_array([net.enc(c)]), ops.ptr_array([self.f_out[c]]), ops.ptr_array([self.f_act[c]]),
                 ops.int_array([R]), H, W, ops.stream())
        else:
            call("tacorl_encoder_fwd", 1, ops.ptr_array([self.frames[c]]), ops.ptr_array([net.enc(c)]),
                 ops.ptr_array([self.f_out[c]]), ops.ptr_array([self.f_act[c]]), ops.int_array([R]), H, W, xd, cd,
                 ops.stream())
        ops.copy_cols(self.f_out[c], 0, 32, self.emb, 32 * j, Ec, R, 32)
    ops.mark("encoded")
    # Branches (torch streams: branches of the captured graph).  The step is a long chain of small kernels; what does not
    # depend on each other runs side by side: the logging-only random-plan decoder pass beside the plan proposal /
    # recognition forward, the plan proposal's backward beside the decoder's, the decoder's and the plan recognition's
    # weight gradients beside the BPTT / input-gradient chains and the encoder backward.  (Scratch buffers are per network - "ad_*" tags - so branches do not share any.)
    if getattr(self, "_branch", None) is None:
        self._branch = [torch.cuda.Stream(device=self.dev) for _ in range(3)]
    main = torch.cuda.current_stream()
    s_rand, s_pp, s_wg = self._branch if getattr(self, "branches", True) else [main] * 3  # tests switch the branches off
    # logging-only pass with a uniform random plan (reference :243-252) - before the real pass, whose activations
    # are the ones the backward sees (the decoder's buffers are shared: the real pass waits for this branch)
    # Where the ring-GEMM path runs (bf16), the random-plan pass is not a pass of its own: its rows ride in the launches of
    # the real pass (same weights, read once - ActionDecoderLogistic.twin_*); only its input projection and its loss
    # stay on the branch.
    twin, ad_prepared, pr_ready = None, False, None
    s_rand.wait_stream(main)
    with torch.cuda.stream(s_rand):
        if cd == BF16 and not getattr(self, "_pr_train", False) and getattr(self, "early_prepare", True):
            # weights only: the plan recognition's bf16 mirror and its composed posterior head, beside the encoder's tail
            pr.prepare_inference()
            pr_ready = torch.cuda.Event()
            pr_ready.record(s_rand)
            # (NOT pr.prepare_backward(): its transposes ride on the weight-gradient stream, behind the decoder's weight
            # gradients - the plan recognition's backward then waits for those, and that serialisation is faster: see there)
        call("tacorl_uniform_actions", ptr(self.noise["u_plan"]), ptr(self.rplan), A, B, A, 0, ops.stream())
        if ad.twin_ok(B, cd):
            twin = ad.twin_input_proj(self.rplan, self.emb, Ec, B, T, T - 1)
            # weights-only preparation of the real pass and of the backward rides on this branch (joined before the pass)
            ad.refresh_mirrors(B, T - 1)
            ad_prepared = ad.prepare_backward(B, T - 1, cd) if getattr(self, "early_prepare", True) else False
            ops.mark("rand:prepared")
        else:
            ad.forward(self.rplan, self.emb, Ec, B, T, T - 1, cd)
            ad.loss(acts, ops._at(self.logs, 4), B, T, T - 1, want_grad=False)
    # The plan proposal's forward (goal encoder -> policy head: a dependent chain of its own, ~55 us at B = 32) needs only the
    # embeddings, like the plan recognition - but on its own branch beside the plan recognition's launch, joined at the KL,
    # the step is SLOWER (1.252 -> 1.302 ms at B = 32, 2.365 -> 2.410 at B = 256, same-process A/B): a third concurrent
    # branch at that point (the early preparation branch is still running) costs more than the 55 us it hides.  In line.
    pp_side = s_pp if getattr(self, "pp_forward_side", False) else main
    pp_side.wait_stream(main)
    with torch.cuda.stream(pp_side):
        head_pp = _playlmp_pp_forward(self, B, T, Ec, A, cd)
    ops.mark("pp_fwd")
    if pr_ready is not None:
        main.wait_event(pr_ready)
    head_pr = pr.forward(self.emb, Ec, B, T, cd, train=getattr(self, "_pr_train", False), prepared=pr_ready is not None,
                         sample=(self.noise["eps_plan"], self.plan) if pr_ready is not None else None)
    main.wait_stream(pp_side)
    pb_g = self._pp_bf16
    call("tacorl_gauss_kl_balanced", ptr(head_pr), ptr(head_pp), ptr(self.d_head_pr), ptr(self.d_head_pp), B, A,
         float(self.kl_alpha), float(self.kl_beta), float(pr.min_std), int(self.kl_balancing), gs, ptr(self.logs),
         ops.stream())
    ops.mark("kl")
    # plan proposal backward (needs only the KL's gradient): its own branch; d_emb gets its share after the join
    s_pp.wait_stream(main)
    with torch.cuda.stream(s_pp):
        def mlp_backward(tag, x, ldx, par, act, d_out, ldo, grad, d_x, ldd, dims, acts):
            if pb_g is not None and ops.mlp_bwd_fused_ok(1, dims, ldo, ldd, cd):
                ops.mlp_bwd_fused_dgrad([par], [act], [d_out], ldo, [d_x], ldd, [B], dims, acts, "plmp_" + tag)
                ops.mlp_bwd_fused_wgrad([x], ldx, [act], [d_out], ldo, [grad], [B], dims, acts, "plmp_" + tag)
            else:
                ops.mlp_bwd([x], ldx, [par], [act], [d_out], ldo, [grad], [d_x], ldd, [B], dims, acts, cd, ws_tag="plmp_" + tag)

        mlp_backward("pp", self.S, 2 * Ec, net.head(), self.pact, self.d_head_pp, 2 * A, net.head(net.grad), self.dS, 2 * Ec,
                     net.head_dims, net.head_acts)
        mlp_backward("genc", self.gin, Ec, net.genc(), self.gact, ops._at(self.dS, Ec), 2 * Ec, net.genc(net.grad), self.dgin, Ec,
                     net.genc_dims, net.genc_acts)
    if pr_ready is None:  # (otherwise pr.forward sampled the plan - inside its one launch where that path runs)
        call("tacorl_pr_sample", ptr(head_pr), ptr(self.noise["eps_plan"]), ptr(self.plan), None, None, B, A,
             float(pr.min_std), ops.stream())
    main.wait_stream(s_rand)
    ops.mark("ad:start")
    # (mirrors_current: the branch above has refreshed the bf16 mirrors of the weights - in the random-plan pass of its own, or explicitly)
    ad.forward(self.plan, self.emb, Ec, B, T, T - 1, cd, mirrors_current=True, twin=twin)
    if twin is not None:
        s_rand.wait_stream(main)
        with torch.cuda.stream(s_rand):
            ad.twin_heads(twin, B, T - 1)  # (the random pass's heads and loss: off the real pass's chain)
            ad.loss(acts, ops._at(self.logs, 4), B, T, T - 1, want_grad=False, twin=twin)
    ad.loss(acts, ops._at(self.logs, 2), B, T, T - 1, want_grad=True, grad_scale=gs)
    if self.add_random_plan_loss:
        raise NotImplementedError("add_random_plan_loss=True is not used by any in-scope config")
    ops.mark("ad:loss")
    # ---- backward
    ad.backward(B, T - 1, cd, need_input_grad=True, wgrad_stream=s_wg, join=False, prepared=ad_prepared)
    if ad.E == Ec:  # (the decoder sees every camera's embedding: its share defines the whole block - no fill launch in front)
        call("tacorl_ad_input_bwd", ptr(ad.dx_seq), ptr(self.d_plan), ptr(self.d_emb), Ec, B, T, T - 1, ad.P, ad.E, 2,
             ops.stream())
    else:
        self.d_emb.zero_()
        call("tacorl_ad_input_bwd", ptr(ad.dx_seq), ptr(self.d_plan), ptr(self.d_emb), Ec, B, T, T - 1, ad.P, ad.E, 1,
             ops.stream())
    ops.mark("ad:bwd")
    call("tacorl_pr_sample_bwd", ptr(head_pr), ptr(self.noise["eps_plan"]), ptr(self.d_plan), ptr(self.d_head_pr), B, A,
         float(pr.min_std), ops.stream())
    dx = pr.backward(self.d_head_pr, B, T, cd, wgrad_stream=s_wg)
    ops.mark("pr:bwd")
    # round 5: the four shares of d_emb and the cameras' slices in ONE launch (three accumulating copies + a copy per camera
    # before, between the plan recognition's backward and the encoders'); demb_one_launch = False: as before
    one = getattr(self, "demb_one_launch", True) and pr.D_in % 4 == 0 and pr.D % 4 == 0 and Ec == 32 * len(cams)
    if one:
        main.wait_stream(s_pp)
        call("tacorl_plmp_demb_finish", ptr(self.d_emb), ptr(dx), pr.D, pr.D_in, ptr(self.dS), 2 * Ec, ptr(self.dgin),
             ops.ptr_array([self.f_dout[c] for c in cams]), len(cams), B, T, Ec, ops.stream())
    else:
        ops.copy_cols(dx, 0, pr.D, self.d_emb, 0, Ec, R, pr.D_in, accumulate=True)
        main.wait_stream(s_pp)
        ops.copy_cols(self.dS, 0, 2 * Ec, self.d_emb, 0, T * Ec, B, Ec, accumulate=True)
        ops.copy_cols(self.dgin, 0, Ec, self.d_emb, (T - 1) * Ec, T * Ec, B, Ec, accumulate=True)
    for j, c in enumerate(cams):
        H, W = hw[c]
        if not one:
            ops.copy_cols(self.d_emb, 32 * j, Ec, self.f_dout[c], 0, 32, R, 32)
        fused = (cd == BF16 and xd == BF16 and bool(ops.L.lib().tacorl_encoder_fused_supported(H, W))
                 and ops.L.lib().tacorl_encoder_bwd_fused_ws_bytes(1, ops.int_array([R]), H, W) > 0)  # forward and backward
        ops.encoder_bwd([self.frames[c]], [net.enc(c)], [self.f_act[c]], [self.f_dout[c]], [net.enc(c, net.grad)], H, W, cd,
                        fused=fused)
    ops.mark("enc:bwd")
    main.wait_stream(s_wg)
    main.wait_stream(s_rand)
    ops.mark("joined")


def _named_gradients(self):
    out = {}
    for k, v in self.net.grad_views.items():
        if k.startswith("encoder."):
            out["perceptual_encoder." + k[len("encoder."):]] = v
        elif k.startswith("actor.policy."):
            out["plan_proposal.policy." + k[len("actor.policy."):]] = v
        else:
            out[k] = v
    out.update({f"plan_recognition.{k}": v for k, v in self.pr.blk.grad_views.items()})
    out.update({f"action_decoder.{k}": v for k, v in self.ad.blk.grad_views.items()})
    return out


def _grad_arena(self):
    """The three blocks' gradients as slices of one allocation (made on first use, i.e. only with several ranks)."""
    a = self.__dict__.get("_grad_arena_t")
    if a is None:
        from ... import ops

        blks = (self.net, self.pr.blk, self.ad.blk)
        ops.note_alloc()
        a = torch.zeros(sum(b.size for b in blks), device=self.dev)
        o = 0
        for b in blks:
            b.rebind_grad(a[o: o + b.size])
            o += b.size
        self.__dict__["_grad_arena_t"] = a
    return a


def _training_step(self, batch, batch_idx=0, noise=None):
    """reference :307-317 (returns the total loss; the optimiser step has already run - manual optimisation)."""
    out = _playlmp_step(self, batch, noise, True, "train")
    self._tick_optimizers()
    return out


def _validation_step(self, batch, batch_idx=0, noise=None):
    return _playlmp_step(self, batch, noise, False, "validation")


def _configure_optimizers(self):
    """reference :362-368: ONE Adam over every parameter.  A BlockAdam spanning the three flat blocks."""
    ent = [(blk, self._pv[k], blk.views_of(blk.m), blk.views_of(blk.v))
           for k, blk in (("net", self.net), ("pr", self.pr.blk), ("ad", self.ad.blk))]
    self._optimizers = [self._make_adam("adam", ent, self.lr)]
    return self._optimizers[0]


PlayLMP.training_step = _training_step
PlayLMP.validation_step = _validation_step
PlayLMP.named_gradients = _named_gradients
PlayLMP.configure_optimizers = _configure_optimizers
