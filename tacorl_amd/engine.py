"""Hand-scheduled actor-critic update (CQL_Offline.compute_update) on HIP kernels.

One `ACEngine.update()` = the reference's `compute_update`
(reference modules/cql/cql_offline_lightning.py:470-542) with the same loss values,
gradients, optimiser steps and soft target update, but scheduled MI355X-first:

* every unique (encoder, image set) is encoded once (11*B images instead of the reference's
  (24+12n)*B): the sample expansion `expand_obs` happens on 64-float embeddings;
* the five networks (actor, q1, q2, target_q1, target_q2) share launches through problem
  batches; every buffer is pre-allocated so the whole update is hipGraph-capturable;
* no autograd: forward and backward are explicit kernel sequences; parameters, gradients,
  Adam moments and targets are flat blocks (clip+Adam+Polyak = 2 launches per network).

Random draws are explicit inputs (`noise` dict, reference draw order - SURVEY 8a note 1).
"""
import contextlib
import os

import torch

from . import blocks, ops
from . import dist as D
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, BF16, F32, LOG_SLOTS, call, ptr


def _al4(x):
    return (x + 3) // 4 * 4


class NetBlock:
    """[encoder(cam) for cam in cams] + goal-encoder MLP + head MLP in one flat buffer."""

    def __init__(self, cams, goal_cams, head_dims, head_acts, head_names, device, head_parts=None, hidden=256):
        self.cams, self.goal_cams = list(cams), list(goal_cams)
        self.all_cams = sorted(set(self.cams) | set(self.goal_cams))
        self.E_obs, self.G = 32 * len(self.cams), 32 * len(self.goal_cams)
        self.genc_dims, self.genc_acts = [self.G, hidden, hidden, self.G], [ACT_RELU, ACT_RELU, ACT_NONE]
        self.head_dims, self.head_acts = list(head_dims), list(head_acts)
        off = 0
        self.enc_off = {}
        for c in self.all_cams:
            self.enc_off[c] = off
            off += blocks.encoder_size()
        self.genc_off = off
        off += blocks.mlp_size(self.genc_dims)
        self.head_off = off
        off += blocks.mlp_size(self.head_dims)
        self.size = off
        z = lambda: torch.zeros(self.size, device=device)  # noqa: E731
        self.param, self.grad, self.m, self.v = z(), z(), z(), z()
        # bf16 copy of the MLP part of the block (the fused MLP forward's MFMA operand); refreshed from
        # the fp32 master at the start of every update (ACEngine._refresh_bf16)
        self.param_bf16 = torch.zeros(self.size, device=device, dtype=torch.bfloat16)
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        self._head_names, self._head_parts = head_names, head_parts
        self.views, self.grad_views = self._views_of(self.param), self._views_of(self.grad)

    def _views_of(self, flat):
        """Reference-named views (logical shapes) into a flat block of this layout."""
        dst = {}
        for c in self.all_cams:
            for k, v in blocks.encoder_views(flat, self.enc_off[c]).items():
                dst[f"encoder.networks.{c}.{k}"] = v
        gn = [(f"goal_encoder.mlp.{i}.weight", f"goal_encoder.mlp.{i}.bias") for i in (0, 2, 4)]
        dst.update(blocks.mlp_views(flat, self.genc_off, self.genc_dims, gn))
        if self._head_parts is None:
            dst.update(blocks.mlp_views(flat, self.head_off, self.head_dims, self._head_names))
        else:
            dst.update(blocks.mlp_views(flat, self.head_off, self.head_dims[:-1], self._head_names))
            dst.update(blocks.head_views(flat, self.head_off, self.head_dims, self.head_dims[-2], self._head_parts))
        return dst

    def views_of(self, flat):
        return self._views_of(flat)

    def rebind_grad(self, flat):
        """Move the gradient block into caller-provided storage (a slice of one arena, so that a single
        all-reduce covers several networks)."""
        assert flat.numel() == self.size and flat.is_contiguous()
        flat.copy_(self.grad)
        self.grad = flat
        self.grad_views = self._views_of(flat)

    def enc(self, cam, flat=None):
        flat = self.param if flat is None else flat
        return flat.data_ptr() + 4 * self.enc_off[cam]

    def genc(self, flat=None):
        return (self.param if flat is None else flat).data_ptr() + 4 * self.genc_off

    def head(self, flat=None):
        return (self.param if flat is None else flat).data_ptr() + 4 * self.head_off

    def genc_bf16(self):
        return self.param_bf16.data_ptr() + 2 * self.genc_off

    def head_bf16(self):
        return self.param_bf16.data_ptr() + 2 * self.head_off


class Scalar:
    def __init__(self, device, value=0.0):
        self.param = torch.full((1,), float(value), device=device)
        self.grad, self.m, self.v = (torch.zeros(1, device=device) for _ in range(3))
        self.step = torch.zeros(1, dtype=torch.int32, device=device)


class ACEngine:
    def __init__(self, cams, goal_cams, hw, action_dim, B, device, *, n=4, discount=0.99, tau=0.005, actor_lr=3e-4,
                 critic_lr=3e-4, deterministic_backup=False, reward_scale=1.0, clip_grad_val=1.0,
                 conservative_weight=1.0, lagrange_thresh=5.0, temp=1.0, with_lagrange=False,
                 discrete_gripper=False, target_entropy=-7.0, policy_layers=3, q_layers=3, hidden=256,
                 compute=F32, img_dtype=torch.float32, world_size=1):
        if sorted(cams) != sorted(goal_cams):
            raise NotImplementedError("observation and goal modalities must coincide (all in-scope configs)")
        self.cams, self.hw, self.B, self.n, self.A = list(cams), dict(hw or {}), B, n, action_dim
        self.dev, self.compute, self.img_dtype = device, compute, img_dtype
        self.dg = bool(discrete_gripper)
        self.Ac = action_dim - 1 if self.dg else action_dim
        self.HD = 2 * self.Ac + (2 if self.dg else 0)
        self.hp = dict(discount=discount, tau=tau, actor_lr=actor_lr, critic_lr=critic_lr,
                       deterministic_backup=deterministic_backup, reward_scale=reward_scale,
                       clip=clip_grad_val, cons_w=conservative_weight, gap=lagrange_thresh, temp=temp,
                       target_entropy=target_entropy)
        self.with_lagrange = with_lagrange
        self.world = world_size
        self.hidden = hidden
        nc = len(self.cams)
        self.Eo = self.G = 32 * nc
        self.E = self.Eo + self.G
        self.ldq = _al4(self.E + action_dim)
        self.lds = self.E
        pol_dims = [self.E] + [hidden] * policy_layers + [self.HD]
        q_dims = [self.E + action_dim] + [hidden] * q_layers + [1]
        silu_p, silu_q = [ACT_SILU] * policy_layers + [ACT_NONE], [ACT_SILU] * q_layers + [ACT_NONE]
        pn = [(f"actor.policy.fc_layers.{i}.weight", f"actor.policy.fc_layers.{i}.bias") for i in range(policy_layers)]
        parts = [("actor.policy.fc_mean", self.Ac), ("actor.policy.fc_log_std", self.Ac)]
        if self.dg:
            parts.append(("actor.policy.gripper_action", 2))
        qn = [(f"critic.Q.fc_layers.{i}.weight", f"critic.Q.fc_layers.{i}.bias") for i in range(q_layers)]
        qn.append(("critic.Q.out.weight", "critic.Q.out.bias"))
        self.actor = NetBlock(cams, goal_cams, pol_dims, silu_p, pn, device, head_parts=parts, hidden=hidden)
        mk = lambda: NetBlock(cams, goal_cams, q_dims, silu_q, qn, device, hidden=hidden)  # noqa: E731
        self.q1, self.q2, self.tq1, self.tq2 = mk(), mk(), mk(), mk()
        self.log_alpha, self.log_alpha_prime = Scalar(device), Scalar(device)
        # one gradient arena [actor | q1 | q2 | log_alpha'] -> ONE all-reduce per step for the update
        # (plus the scalar one for log_alpha, which must be stepped before the actor loss)
        self.arena_extra = []
        self._bind_arena()
        self.extra_enc, self._wpk = [], {}
        self.B = None
        if B:
            self.ensure_batch(B)

    def _bind_arena(self):
        """[actor | q1 | q2 | log_alpha' | extra blocks...]: every gradient the step's second collective reduces, one
        allocation (the fine-tuned action decoder's block rides in the same all-reduce)."""
        blks = [self.actor, self.q1, self.q2]
        sizes = [b.size for b in blks] + [4] + [b.size for b in self.arena_extra]
        arena = torch.zeros(sum(sizes), device=self.dev)
        o = 0
        for blk in blks:
            blk.rebind_grad(arena[o: o + blk.size])
            o += blk.size
        old = self.log_alpha_prime.grad
        arena[o: o + 1].copy_(old.reshape(-1)[:1])
        self.log_alpha_prime.grad = arena[o: o + 1]
        o += 4
        for blk in self.arena_extra:
            blk.rebind_grad(arena[o: o + blk.size])
            o += blk.size
        self.grad_arena = arena

    def extend_arena(self, blk):
        """Put another trainable block's gradients (TACORL: the fine-tuned action decoder) into the arena."""
        if not any(b is blk for b in self.arena_extra):
            ops.note_alloc()
            self.arena_extra.append(blk)
            self._bind_arena()

    # ------------------------------------------------------------------ buffers
    def ensure_batch(self, B, hw=None):
        """(Re)allocate every batch-sized buffer; parameters and optimiser state are untouched."""
        hw = dict(hw) if hw is not None else self.hw
        if self.B != B or hw != self.hw:
            self.B, self.hw = B, hw
            self._alloc()

    def _alloc(self):
        B, n, dev = self.B, self.n, self.dev
        ops.note_alloc()
        f = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
        self.X3 = {c: torch.zeros(3 * B, *self.hw[c], 3, device=dev, dtype=self.img_dtype) for c in self.cams}
        # encoder problems: (net, first image row in X3, n images, keep activations for backward)
        self.enc_probs = [("a_og", self.actor, 0, 2 * B), ("a_nx", self.actor, 2 * B, B), ("q1", self.q1, 0, 2 * B),
                          ("q2", self.q2, 0, 2 * B), ("tq1", self.tq1, B, 2 * B), ("tq2", self.tq2, B, 2 * B)]
        self.enc_out = {(k, c): f(nimg, 32) for k, _, _, nimg in self.enc_probs for c in self.cams}
        self.enc_act = {(k, c): f(ops.encoder_act_layout(nimg, *self.hw[c])[1])
                        for k, _, _, nimg in self.enc_probs for c in self.cams}
        self.enc_dout = {(k, c): f(2 * B, 32) for k in ("a_og", "q1", "q2") for c in self.cams}
        nets = [("a", self.actor), ("q1", self.q1), ("q2", self.q2), ("tq1", self.tq1), ("tq2", self.tq2)]
        self.nets = nets
        self.gin = {k: f(B, self.G) for k, _ in nets}
        self.gact = {k: f(ops.mlp_act_layout(B, net.genc_dims, net.genc_acts)[2]) for k, net in nets}
        self.g_yoff = ops.mlp_act_layout(B, self.actor.genc_dims, self.actor.genc_acts)[1][-1]
        self.S = {k: f(B, self.lds) for k in ("a", "a_nx", "q1", "q2", "tq1", "tq2")}
        pd, pa = self.actor.head_dims, self.actor.head_acts
        self.pact = {k: f(ops.mlp_act_layout(B, pd, pa)[2]) for k in ("a", "a_nx")}
        self.p_yoff = ops.mlp_act_layout(B, pd, pa)[1][-1]
        R = (3 * n + 1) * B
        self.R = R
        self.acts_main, self.act_pi, self.act_next = f(R, self.A), f(B, self.A), f(B, self.A)
        self.logp_pi, self.logp_next = f(B), f(B)
        self.logp_cur, self.logp_nxt = f(n * B), f(n * B)
        self.grip_pi = torch.zeros(B, dtype=torch.int32, device=dev)
        qd, qa = self.q1.head_dims, self.q1.head_acts
        self.XQ = {k: f(R, self.ldq) for k in ("q1", "q2")}
        self.XQpi = {k: f(B, self.ldq) for k in ("q1", "q2")}
        self.XT = {k: f(B, self.ldq) for k in ("tq1", "tq2")}
        self.qact = {k: f(ops.mlp_act_layout(R, qd, qa)[2]) for k in ("q1", "q2")}
        self.qact_pi = {k: f(ops.mlp_act_layout(B, qd, qa)[2]) for k in ("q1", "q2")}
        self.qact_t = {k: f(ops.mlp_act_layout(B, qd, qa)[2]) for k in ("tq1", "tq2")}
        self.q_yoff_R = ops.mlp_act_layout(R, qd, qa)[1][-1]
        self.q_yoff_B = ops.mlp_act_layout(B, qd, qa)[1][-1]
        self.dq = {k: f(R) for k in ("q1", "q2")}
        self.dq_pi = {k: f(B) for k in ("q1", "q2")}
        self.dXQ = {k: f(R, self.ldq) for k in ("q1", "q2")}
        self.dXQpi = {k: f(B, self.ldq) for k in ("q1", "q2")}
        self.d_head = f(B, self.HD)
        self.dS = {k: f(B, self.lds) for k in ("a", "q1", "q2")}
        self.dgin = {k: f(B, self.G) for k in ("a", "q1", "q2")}
        self.reward, self.done = f(B), f(B)
        # the data action IS rows [0, B) of the Q networks' action column block (no copy in front of the Q forward):
        # load_transition / TACORL's plan-recognition sample write it in place
        self.action = self.acts_main[:B]
        # all noise of a step in two flat buffers (normal | uniform): two generator launches per step
        shapes = dict(eps_pi=(B, self.Ac), eps_next=(B, self.Ac), eps_cur=(n, B, self.Ac), eps_nxt=(n, B, self.Ac))
        shapes.update(getattr(self, "extra_normal", {}))  # e.g. TACORL's plan-recognition eps
        ushapes = dict(u_rand=(n * B, self.A))
        if self.dg:
            ushapes.update(g_pi=(B, 2), g_next=(B, 2), g_cur=(n, B, 2), g_nxt=(n, B, 2))

        def carve(sh):
            import math
            sizes = {k: _al4(math.prod(v)) for k, v in sh.items()}
            flat = f(sum(sizes.values()))
            out, off = {}, 0
            for k, v in sh.items():
                out[k] = flat[off: off + math.prod(v)].view(*v)
                off += sizes[k]
            return flat, out

        self._noise_normal, nn_ = carve(shapes)
        self._noise_uniform, nu_ = carve(ushapes)
        self.extra_noise = {k: nn_.pop(k) for k in getattr(self, "extra_normal", {})}
        self.noise = dict(eps_pi=nn_["eps_pi"], eps_next=nn_["eps_next"], u_rand=nu_["u_rand"], eps_cur=nn_["eps_cur"],
                          eps_nxt=nn_["eps_nxt"])
        if self.dg:
            self.noise.update({k: nu_[k] for k in ("g_pi", "g_next", "g_cur", "g_nxt")})
        self.logs = f(32)
        self.cql_ws = torch.empty(max(256, ops.L.lib().tacorl_cql_ws_bytes(B)), dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------- inputs
    def set_noise(self, noise=None):
        """Copy injected noise, or draw fresh noise with torch's device generator (two launches)."""
        if noise is None:
            self._noise_normal.normal_()
            self._noise_uniform.uniform_()
            return
        for k, buf in self.noise.items():
            buf.copy_(noise[k].reshape(buf.shape))
        for k, buf in self.extra_noise.items():
            if k in noise:
                buf.copy_(noise[k].reshape(buf.shape))

    def load_images(self, cam, obs, goal, nxt, nchw=True):
        """obs/goal/nxt: (B,3,H,W) [nchw] or (B,H,W,3) fp32 device tensors (may be strided views with a
        uniform image pitch, e.g. states[:,0])."""
        H, W = self.hw[cam]
        xd = BF16 if self.img_dtype == torch.bfloat16 else F32
        esz, img = self.X3[cam].element_size(), H * W * 3
        jobs = []
        if obs.dtype == torch.uint8:
            # the dataset's uint8 HWC frames: ToTensor + Normalize(0.5, 0.5) applied by the pack (bit-identical to the
            # host-transformed fp32 route, a quarter of the bytes)
            for i, t in enumerate((obs, goal, nxt)):
                assert t.is_cuda and t.dtype == torch.uint8 and t[0].is_contiguous() and tuple(t.shape[-3:]) == (H, W, 3)
                pitch = t.stride(0) if t.shape[0] > 1 else img
                jobs.append((t.data_ptr(), pitch, self.X3[cam].data_ptr() + i * self.B * img * esz, self.B))
            if img % 16 or any(j[0] % 16 or j[1] % 16 for j in jobs):
                raise ValueError("uint8 frames: H*W*3 and the image pitch must be multiples of 16, tensors 16-byte aligned")
            ops.pack_images_u8_batch(jobs, xd, H, W)
            return
        for i, t in enumerate((obs, goal, nxt)):
            assert t.is_cuda and t.dtype == torch.float32 and t[0].is_contiguous()
            pitch = t.stride(0) if t.shape[0] > 1 else 3 * H * W
            jobs.append((t.data_ptr(), pitch, self.X3[cam].data_ptr() + i * self.B * img * esz, self.B))
        if nchw and (H * W) % 4 == 0 and all(j[0] % 16 == 0 and j[1] % 4 == 0 for j in jobs):
            ops.pack_images_batch(jobs, xd, H, W)  # one vectorised launch for obs / goal / next
        else:
            for src, pitch, dst, n_ in jobs:
                call("tacorl_pack_images", src, pitch, int(nchw), dst, xd, n_, 3, H, W, ops.stream())

    def load_transition(self, action, reward, done):
        self.action.copy_(action.reshape(self.B, self.A).float())
        self.reward.copy_(reward.reshape(self.B).float())
        self.done.copy_(done.reshape(self.B).float())

    # ----------------------------------------------------------------- forward
    def _img_ptr(self, cam, first_row):
        H, W = self.hw[cam]
        return self.X3[cam].data_ptr() + first_row * H * W * 3 * self.X3[cam].element_size()

    # Every encoder problem of a camera goes through the fused single-launch kernel when it applies
    # (bf16 images + bf16 MFMA + a templated camera geometry); the ones a backward follows (GRAD_PROBS)
    # also get their activations saved by that launch.  Otherwise: the per-layer path.
    GRAD_PROBS = ("a_og", "q1", "q2")
    use_fused = True  # tests flip this to compare the fused launch against the per-layer path
    # MLP weight gradients on side streams (parallel graph branches): they are needed only by Adam, so
    # they leave the dependent chain of the update (1.70 -> 1.60 ms/step at the bench shapes)
    # MLP weight gradients on side streams (graph branches joined before the optimiser) or in line behind their input-gradient
    # launch: True / False / a collection of site tags ("q", "pi", "genc").  In line since round 3: beside the action-decoder
    # branch's chip-wide GEMMs a third set of concurrent launches costs the chain more than its own kernels' time
    # (same-process A/B on the headline step: 0.8801 -> 0.8682 ms; TACORL_WGRAD_SIDE=1 restores the branches)
    wgrad_side_streams = os.environ.get("TACORL_WGRAD_SIDE", "0") == "1"
    conv_wgrad_side_stream = False  # measured: 0.97 -> 1.12 ms/step when on (co-resident workgroups slow the chain); conv3 / conv2 weight gradients beside the dgrad chain (see _encoders_backward)

    def _fused_ok(self, c):
        if not self.use_fused or self.compute != BF16 or self.img_dtype != torch.bfloat16:
            return False
        return bool(ops.L.lib().tacorl_encoder_fused_supported(*self.hw[c]))

    def _fused_bwd_ok(self, c):
        """The per-image LDS-resident conv backward may exist for fewer geometries than the fused forward; where it does
        not, the problems that have a backward take the per-layer forward (fp32 activations)."""
        return self._fused_ok(c) and ops.L.lib().tacorl_encoder_bwd_fused_ws_bytes(
            3, ops.int_array([2 * self.B] * 3), *self.hw[c]) > 0

    def _packed(self, net, c):
        key = (id(net), c)
        if key not in self._wpk:
            ops.note_alloc()
            self._wpk[key] = torch.empty(ops.L.lib().tacorl_encoder_fused_wpk_bytes(), dtype=torch.uint8, device=self.dev)
        return self._wpk[key]

    def _all_problems(self, c, which="all"):
        """(image pointer, net, out, act, n_img, needs_backward, camera) of every encoder problem of camera c
        (which = "own": the update's networks only, "extra": the caller's frozen ones only - TACORL's LMP window)."""
        pr = [] if which == "extra" else [
            (self._img_ptr(c, r0), net, self.enc_out[(k, c)], self.enc_act[(k, c)], n, k in self.GRAD_PROBS, c)
            for k, net, r0, n in self.enc_probs]
        if which != "own":
            pr += [(x["img"], x["net"], x["out"], x["act"], x["n"], False, c) for x in self.extra_enc if x["cam"] == c]
        return pr

    def encode_split(self, between):
        """(TACORL_EF_SPLIT_LMP, fused path only): the frozen extra problems (TACORL: the LMP window, whose
        embeddings the plan recognition -> action decoder branch waits for) as a launch of their own FIRST, `between()` (the
        caller forks that branch there), then the update's own problems on TACORL_EF_SPLIT_BUDGET workgroups (default 160:
        96 CUs stay free for the branch; sweep 96 .. 240 on C4 / C3: 128 - 160 best).  Returns False when the split does not apply (nothing launched)."""
        groups = self._fused_groups()
        if not self.extra_enc or len(groups) != 1 or sorted(groups[0]) != sorted(self.cams) or not all(self._fused_bwd_ok(c) for c in self.cams):
            return False
        cs = groups[0]
        for which in ("extra", "own"):
            pr = [x for c in cs for x in self._all_problems(c, which)]
            for c in cs:
                self._pack_encoders(c, list({id(x[1]): x[1] for x in pr if x[6] == c}.values()), only_stale=self.ef_pack_late)
            if which == "own":
                self._launch_fused(cs[0], pr, max_wg=int(os.environ.get("TACORL_EF_SPLIT_BUDGET", "160")))
            else:
                self._launch_fused(cs[0], pr)
                between()
        return True

    # Packed conv weights of the fused encoder forward (bf16 MFMA fragments in the kernel's register order).  Round 5: the
    # pack launch of the networks the optimiser moves runs BEHIND the Adam launch, at the end of the step - in the shadow of
    # the action-decoder branch, which ends later - instead of in front of the encoder forward at the head of the next
    # step's chain; a packed copy counts as current while the block's torch version counter is the one recorded when it was
    # packed (see mirrors_stale), so frozen networks (TACORL's LMP encoder) are packed once, not every step.
    # TACORL_EF_PACK_LATE=0: every network in front of every forward, as before.
    ef_pack_late = os.environ.get("TACORL_EF_PACK_LATE", "1") == "1"

    def _pack_encoders(self, c, nets, only_stale=False):
        vers = self.__dict__.setdefault("_wpk_ver", {})
        if only_stale:
            nets = [n_ for n_ in nets if vers.get((id(n_), c)) != n_.param._version]
        if not nets:
            return
        call("tacorl_encoder_pack_weights", len(nets), ops.ptr_array([n_.enc(c) for n_ in nets]),
             ops.ptr_array([self._packed(n_, c) for n_ in nets]), ops.stream())
        for n_ in nets:
            vers[(id(n_), c)] = n_.param._version

    def _late_pack_nets(self):
        return [self.actor, self.q1, self.q2, self.tq1, self.tq2]

    def packs_stale(self):
        """Would a replayed step read a packed copy that no longer matches its parameter block?  (Every network of the
        step's encoder launch: the optimiser's ones AND the extra, frozen ones.)"""
        if not self.ef_pack_late or not getattr(self, "_wpk_ver", None):
            return False
        nets = {id(n_): n_ for n_ in self._late_pack_nets()}
        nets.update({id(x["net"]): x["net"] for x in self.extra_enc})
        return any(self._wpk_ver.get((i, c), n_.param._version) != n_.param._version
                   for i, n_ in nets.items() for c in self.cams if (i, c) in self._wpk_ver)

    def packs_written(self):
        """The step's tail launch (eager, or the replayed graph's) has just re-packed the optimiser's networks: record it."""
        vers = getattr(self, "_wpk_ver", None)
        if self.ef_pack_late and vers:
            for n_ in self._late_pack_nets():
                for c in self.cams:
                    if (id(n_), c) in vers:
                        vers[(id(n_), c)] = n_.param._version

    def _launch_fused(self, c, pr, max_wg=0):
        """One fused encoder launch over the problems pr (each carries its camera, x[6]: cameras of one geometry may share a
        launch) on at most max_wg workgroups (0: one per CU)."""
        H, W = self.hw[c]
        call("tacorl_encoder_fwd_fused_wg", len(pr), ops.ptr_array([x[0] for x in pr]),
             ops.ptr_array([self._packed(x[1], x[6]) for x in pr]), ops.ptr_array([x[1].enc(x[6]) for x in pr]),
             ops.ptr_array([x[2] for x in pr]), ops.ptr_array([x[3] if x[5] else None for x in pr]),
             ops.int_array([x[4] for x in pr]), H, W, int(max_wg), ops.stream())

    def _fused_groups(self):
        """Cameras whose fused encoder problems share ONE launch: the cameras of one geometry when their problems fit the
        launch's table (round 5; C4: two 128 x 128 cameras, 7 problems each - one launch over 5 504 images instead of two over
        2 752: one prologue, one tail); every other fused camera alone.  TACORL_EF_MERGE_CAMS=0: one launch per camera."""
        groups, out = {}, []
        for c in self.cams:
            if self._fused_ok(c):
                groups.setdefault((tuple(self.hw[c]), bool(self._fused_bwd_ok(c))), []).append(c)
        for (_, bwd_ok), cs in groups.items():
            if (bwd_ok and len(cs) > 1 and os.environ.get("TACORL_EF_MERGE_CAMS", "1") == "1"
                    and sum(len(self._all_problems(c)) for c in cs) <= 16):
                out.append(cs)
            else:
                out += [[c] for c in cs]
        return out

    def encode_fused_only(self):
        """Just the fused encoder launch(es) of the step (bench roofline probe).  Returns images per step."""
        n_total = 0
        for cs in self._fused_groups():
            pr = [x for c in cs for x in self._all_problems(c)]
            self._launch_fused(cs[0], pr)
            n_total += sum(x[4] for x in pr)
        return n_total

    def _encode_all(self):
        """Every encoder forward of the step: ONE fused launch per camera (27*B images at TACORL shapes:
        frozen LMP window, actor(obs, goal, next), q1, q2 and both targets), activations saved only for the
        problems that have a backward; the per-layer path covers fp32 mode / images too large for LDS."""
        xd = BF16 if self.img_dtype == torch.bfloat16 else F32
        merged = {c: cs for cs in self._fused_groups() if len(cs) > 1 for c in cs}
        for c in self.cams:
            H, W = self.hw[c]
            pr = self._all_problems(c)
            if c in merged:
                if merged[c][0] != c:
                    continue  # (launched with the group's first camera)
                allp = []
                for cc in merged[c]:
                    prc = self._all_problems(cc)
                    self._pack_encoders(cc, list({id(x[1]): x[1] for x in prc}.values()), only_stale=self.ef_pack_late)
                    allp += prc
                self._launch_fused(c, allp)
                continue
            if self._fused_ok(c):
                # (act format 2 - a geometry of encoder_ring.hip without the LDS-resident backward: the fused launch saves fp32
                # activations, which the per-layer backward reads)
                saves = self._fused_bwd_ok(c) or ops.L.lib().tacorl_encoder_fused_act_format(H, W) == 2
                slow = [] if saves else [x for x in pr if x[5]]
                pr = [x for x in pr if not (slow and x[5])]
                nets = {id(x[1]): x[1] for x in pr}
                self._pack_encoders(c, list(nets.values()), only_stale=self.ef_pack_late)
                self._launch_fused(c, pr)
                pr = slow
            if pr:
                call("tacorl_encoder_fwd", len(pr), ops.ptr_array([x[0] for x in pr]),
                     ops.ptr_array([x[1].enc(c) for x in pr]), ops.ptr_array([x[2] for x in pr]),
                     ops.ptr_array([x[3] for x in pr]), ops.int_array([x[4] for x in pr]), H, W, xd, self.compute,
                     ops.stream())

    # bf16 mirrors of the MLP weights (the fused MLP kernels' MFMA operand).  Round 5: the Adam / Polyak launch writes them
    # itself (ops.adam_step_batch(mirrors=...)), so the conversion launch at the head of the next step's chain is gone; a
    # mirror counts as current while its block's torch version counter is the one recorded when it was written (every
    # other writer - load_state_dict, broadcast, sync_targets, a user's in-place edit - moves the counter), and the
    # conversion launch below runs only for blocks that are not (first step, after a load).
    # MEASURED SLOWER, default off (round 5, separate processes on one box, ms/step: 0.8370 with the conversion launch
    # against 0.8431 - 0.8485 without it): the plan recognition's 256 one-per-CU workgroups (351 registers: no fused-MLP
    # workgroup fits beside one) start their launch beside the first kernels of phase_a; whatever delays phase_a's first
    # MLP launch by a few microseconds lets all of them start at once, and the action-decoder branch behind them - the
    # step's co-critical chain - starts that much earlier.  The conversion launch is such a delay and costs nothing else.
    adam_writes_mirrors = os.environ.get("TACORL_ADAM_MIRRORS", "0") == "1"

    def _mirror_nets(self):
        return [self.actor, self.q1, self.q2, self.tq1, self.tq2]

    def mirrors_stale(self):
        return self.compute == BF16 and any(getattr(n_, "_mirror_ver", None) != n_.param._version for n_ in self._mirror_nets())

    def mirrors_written(self):
        """The optimiser launch (eager, or the replayed graph's) has just written every mirror: record the versions."""
        if self.compute == BF16 and self.adam_writes_mirrors:
            for n_ in self._mirror_nets():
                n_._mirror_ver = n_.param._version

    def _refresh_bf16(self):
        """bf16 copies of the five networks' MLP weights (one launch) where they are not current; the fp32 blocks stay the masters."""
        if self.compute != BF16:
            return
        nets = [n_ for n_ in self._mirror_nets()
                if not self.adam_writes_mirrors or getattr(n_, "_mirror_ver", None) != n_.param._version
                or os.environ.get("TACORL_ALWAYS_REFRESH") == "1"]
        if not nets:
            return
        call("tacorl_to_bf16_batch", len(nets), ops.ptr_array([n_.genc() for n_ in nets]),
             ops.ptr_array([n_.genc_bf16() for n_ in nets]), (ops.C.c_long * len(nets))(*[n_.size - n_.genc_off for n_ in nets]),
             ops.stream())
        for n_ in nets:
            n_._mirror_ver = n_.param._version

    def _mlp_bwd_sites(self):
        """(tag, params, M, dims) of every fused-MLP backward of the update, as _mlp_backward receives them."""
        B = self.B
        qd, pd, gd = self.q1.head_dims, self.actor.head_dims, self.actor.genc_dims
        return [("q", [self.q1.head(), self.q2.head()], [self.R, self.R], qd),
                ("qpi", [self.q1.head(), self.q2.head()], [B, B], qd),
                ("pi", [self.actor.head()], [B], pd),
                ("genc", [self.actor.genc(), self.q1.genc(), self.q2.genc()], [B] * 3, gd)]

    def _prepack_backward(self):
        """Weight-only preparation of the backward kernels (transposed MLP weights, conv W^T fragments) on a
        side stream while the forward runs: six small launches that would otherwise sit on the dependent
        chain of the backward."""
        self._prepacked = False
        if self.compute != BF16:
            return
        # in line since round 5 (TACORL_PACK_INLINE=0: a side branch of the graph, as before): these launches are small
        # enough in registers to run beside the plan recognition's workgroups, which the fused MLP forwards that follow
        # are not, and without the fork / join the step is 8 us shorter (0.8467 / 0.8432 -> 0.8370 ms, same box)
        inline = os.environ.get("TACORL_PACK_INLINE", "1") == "1"
        if getattr(self, "_pack_stream", None) is None:
            self._pack_stream = torch.cuda.Stream(device=self.dev)
        ps = torch.cuda.current_stream() if inline else self._pack_stream
        if not inline:
            ps.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ps):
            for tag, params, M, dims in self._mlp_bwd_sites():
                ops.mlp_bwd_fused_pack(params, M, dims, "mlp_bwdf_" + tag, self.dev)
            nets = [self.actor, self.q1, self.q2]
            for cs in self._ebw_sequences():
                c = cs[0]
                if self._fused_bwd_ok(c):
                    H, W = self.hw[c]
                    np_ = 3 * len(cs)
                    n3 = ops.int_array([2 * self.B] * np_)
                    nb = ops.L.lib().tacorl_encoder_bwd_fused_ws_bytes(np_, n3, H, W)
                    ws = ops.workspace(nb, self.dev, "enc_bwd_fused_" + "+".join(cs))
                    call("tacorl_encoder_bwd_fused_pack", np_, ops.ptr_array([x.enc(cc) for cc in cs for x in nets]), n3, H, W, ptr(ws),
                         ws.numel(), ops.stream())
        self._prepacked = True

    # Gathered MLP inputs (round 5): the concatenations in front of the goal encoders, the policy head and the Q heads are
    # read by the fused forward's input stage where their producers left them (ops.mlp_fwd_gather) instead of being
    # assembled by a copy launch first - three launches fewer on the step's dependent chain.  TACORL_MLP_GATHER=0 (or a
    # shape the gathered forward does not take: f32 mode, C5's many-row Q problems) restores copy + forward.
    gather_inputs = os.environ.get("TACORL_MLP_GATHER", "1") == "1"

    def _gather_ok(self, tag):
        if not (self.gather_inputs if self.gather_inputs is not None else os.environ.get("TACORL_MLP_GATHER", "1") == "1") \
                or self.compute != BF16:
            return False
        cache = self.__dict__.setdefault("_gather_cache", {})
        key = (tag, self.B)
        if key not in cache:
            B = self.B
            M, net, ldx = {"genc": ([B] * 5, (self.actor.genc_dims, self.actor.genc_acts), self.G),
                           "pi": ([B] * 2, (self.actor.head_dims, self.actor.head_acts), self.lds),
                           "q": ([self.R, self.R, B, B, B, B], (self.q1.head_dims, self.q1.head_acts), self.ldq)}[tag]
            # (segment pitches must be multiples of 4 floats: the action block's is A - 7 for the CQL baseline; every
            # segment starts on a multiple of 8 columns at a 16-byte aligned address - mlp_set_gather's conditions, checked
            # here so that a geometry outside them takes the copy + forward path instead of failing inside the step)
            geom_ok = self.G % 4 == 0 and self.Eo % 8 == 0 and self.E % 8 == 0 and self.g_yoff % 4 == 0
            cache[key] = geom_ok and (tag != "q" or self.A % 4 == 0) and ops.mlp_fwd_gather_ok(
                M, net[0], net[1], ldx, self.compute, self._lean(tag))
        return cache[key]

    def _emb_segs(self, ek, row0, gk, mod=0):
        """[enc(obs or next) per camera | goal_enc(enc(goal))] as input segments."""
        segs = [(self.enc_out[(ek, c)], row0 * 32, 32, 32 * j, mod) for j, c in enumerate(self.cams)]
        segs.append((self.gact[gk], self.g_yoff, self.G, self.Eo, mod))
        return segs

    _OBS_SRC = {"a": ("a_og", 0, "a"), "a_nx": ("a_nx", 0, "a"), "q1": ("q1", 0, "q1"), "q2": ("q2", 0, "q2"),
                "tq1": ("tq1", 1, "tq1"), "tq2": ("tq2", 1, "tq2")}  # (encoder problem, first row / B, goal-encoder net)

    def _assemble_states(self):
        B = self.B
        # goal-encoder inputs: concat over cams of enc(goal)
        src = {"a": ("a_og", B), "q1": ("q1", B), "q2": ("q2", B), "tq1": ("tq1", 0), "tq2": ("tq2", 0)}
        nets = dict(self.nets)
        ks = ["a", "q1", "q2", "tq1", "tq2"]
        if self._gather_ok("genc"):
            segs = [[(self.enc_out[(src[k][0], c)], src[k][1] * 32, 32, 32 * j, 0) for j, c in enumerate(self.cams)] for k in ks]
            ops.mlp_fwd_gather(segs, [self.gin[k] for k in ks], self.G, [nets[k].genc() for k in ks],
                               [nets[k].genc_bf16() for k in ks], [self.gact[k] for k in ks], [B] * 5, self.actor.genc_dims,
                               self.actor.genc_acts, lean=self._lean("genc"))
        else:
            with ops.copy_batch():
                for k, (ek, row0) in src.items():
                    for j, c in enumerate(self.cams):
                        ops.copy_cols(self.enc_out[(ek, c)], row0 * 32, 32, self.gin[k], 32 * j, self.G, B, 32)
            ops.mlp_fwd([self.gin[k] for k in ks], self.G, [nets[k].genc() for k in ks], [self.gact[k] for k in ks],
                        [B] * 5, self.actor.genc_dims, self.actor.genc_acts, self.compute,
                        params_bf16=[nets[k].genc_bf16() for k in ks], lean=self._lean("genc"))
        # S = [enc(obs or next) | goal_enc(enc(goal))]: assembled by a copy only for the forwards that do not gather
        need = [k for k in self._OBS_SRC if not self._gather_ok("pi" if k in ("a", "a_nx") else "q")]
        if need:
            with ops.copy_batch():
                for k in need:
                    ek, r0, gk = self._OBS_SRC[k]
                    for j, c in enumerate(self.cams):
                        ops.copy_cols(self.enc_out[(ek, c)], r0 * B * 32, 32, self.S[k], 32 * j, self.lds, B, 32)
                    ops.copy_cols(self.gact[gk], self.g_yoff, self.G, self.S[k], self.Eo, self.lds, B, self.G)

    def _policy_fwd(self):
        ks = ["a", "a_nx"]
        if self._gather_ok("pi"):
            B = self.B
            segs = [self._emb_segs(self._OBS_SRC[k][0], self._OBS_SRC[k][1] * B, self._OBS_SRC[k][2]) for k in ks]
            # (S["a"] is the policy head's layer-0 operand in the backward: written from the forward's registers)
            ops.mlp_fwd_gather(segs, [self.S["a"], None], self.lds, [self.actor.head()] * 2, [self.actor.head_bf16()] * 2,
                               [self.pact[k] for k in ks], [B] * 2, self.actor.head_dims, self.actor.head_acts,
                               lean=self._lean("pi"))
            return
        ops.mlp_fwd([self.S[k] for k in ks], self.lds, [self.actor.head()] * 2, [self.pact[k] for k in ks],
                    [self.B] * 2, self.actor.head_dims, self.actor.head_acts, self.compute,
                    params_bf16=[self.actor.head_bf16()] * 2, lean=self._lean("pi"))

    def _head(self, k):
        return self.pact[k][self.p_yoff: self.p_yoff + self.B * self.HD]

    def update(self, bc_phase, optimize=True, encoded=False):
        """One compute_update.  Inputs must have been staged with load_images / load_transition /
        set_noise.  Returns nothing; metrics are in self.logs (read with metrics()).
        encoded=True: the caller already ran _encode_all() for this batch.

        Three collective-free phases (each hipGraph-capturable) separated by the step's two all-reduces:
        a) forward up to the alpha gradient; b) alpha step, critics, losses, every backward;
        c) optimiser steps + soft target update."""
        self.phase_a(encoded, optimize)
        self.allreduce_alpha()
        self.phase_b(bc_phase, optimize)
        self.allreduce_grads()
        self.phase_c(optimize)

    def allreduce_alpha(self):
        self._allreduce([self.log_alpha.grad])

    def allreduce_grads(self):
        self._allreduce([self.grad_arena])

    def phase_a(self, encoded=False, optimize=True):
        B, n, A, Ac, hp, nz = self.B, self.n, self.A, self.Ac, self.hp, self.noise
        gs = 1.0 / self.world
        if not encoded:
            self._encode_all()
        ops.mark("a:start")
        # (round 5: the weight-only launches of the two calls below - bf16 mirrors, transposed weights of the four MLP backward
        # sites and of the encoders' FC tails: 7 launches of 5 - 7 us on this chain - leave as ONE, ops.prep_batch)
        with ops.prep_batch() if os.environ.get("TACORL_PACK_INLINE", "1") == "1" else contextlib.nullcontext():
            self._refresh_bf16()
            self._prepack_backward()
        self._assemble_states()
        self._policy_fwd()
        ops.mark("a:policy_fwd")
        head_cur, head_next = self._head("a"), self._head("a_nx")
        g = (lambda k: nz[k]) if self.dg else (lambda k: None)
        # actor rsample on obs, critic-target sample on next_obs, CQL samples on both
        # one launch: rsample on obs, critic-target sample on next_obs, the n CQL samples on both, uniform actions
        at = ops._at
        jobs = [(head_cur, nz["eps_pi"], g("g_pi"), 1, ptr(self.act_pi), self.logp_pi, self.grip_pi if self.dg else None, 1),
                (head_next, nz["eps_next"], g("g_next"), 0, ptr(self.act_next), self.logp_next, None, 1),
                (head_cur, nz["eps_cur"], g("g_cur"), 0, at(self.acts_main, (1 + n) * B * A), self.logp_cur, None, n),
                (head_next, nz["eps_nxt"], g("g_nxt"), 0, at(self.acts_main, (1 + 2 * n) * B * A), self.logp_nxt, None, n)]
        call("tacorl_tanh_normal_sample_batch", len(jobs), ops.ptr_array([j[0] for j in jobs]), self.HD,
             ops.ptr_array([j[1] for j in jobs]), ops.ptr_array([j[2] for j in jobs]), ops.int_array([j[3] for j in jobs]),
             ops.ptr_array([j[4] for j in jobs]), A, ops.ptr_array([j[5] for j in jobs]),
             ops.ptr_array([j[6] for j in jobs]), ops.int_array([j[7] for j in jobs]), B, Ac, ptr(nz["u_rand"]),
             at(self.acts_main, B * A), n * B, A, int(self.dg), ops.stream())
        # alpha: loss, gradient, Adam step (alpha is read post-step below; SURVEY 8a note 2)
        # (one GPU and an optimising step: loss + gradient + Adam in one launch; otherwise the step follows collective #1)
        self._alpha_stepped = bool(optimize and not D.collectives_on(self.world) and not getattr(self, "split_alpha_step", False))
        if self._alpha_stepped:
            la = self.log_alpha
            call("tacorl_alpha_loss_step", ptr(self.logp_pi), B, ptr(la.param), float(hp["target_entropy"]), ptr(la.grad),
                 ptr(self.logs), ptr(la.m), ptr(la.v), float(hp["actor_lr"]), ptr(la.step), ops.stream())
        else:
            call("tacorl_alpha_loss", ptr(self.logp_pi), B, ptr(self.log_alpha.param), float(hp["target_entropy"]), gs,
                 ptr(self.log_alpha.grad), ptr(self.logs), ops.stream())
        ops.mark("a:alpha")
        if getattr(self, "_prepacked", False) and os.environ.get("TACORL_PACK_INLINE", "1") != "1":  # the side branch ends inside this phase
            torch.cuda.current_stream().wait_stream(self._pack_stream)
        ops.mark("a:end")

    def phase_b(self, bc_phase, optimize=True):
        B, n, A, Ac, hp, nz = self.B, self.n, self.A, self.Ac, self.hp, self.noise
        gs = 1.0 / self.world
        head_cur = self._head("a")
        if optimize and not getattr(self, "_alpha_stepped", False):
            ops.adam_step(self.log_alpha.param, self.log_alpha.grad, self.log_alpha.m, self.log_alpha.v,
                          hp["actor_lr"], 0.0, self.log_alpha.step)
        # Q inputs: [S | action]; the data action is read here first (phase_a does not need it, so a caller
        # may still be producing it - TACORL's plan recognition runs beside phase_a): wait for it if asked
        if getattr(self, "action_ready", None) is not None:
            torch.cuda.current_stream().wait_event(self.action_ready)
            self.action_ready = None
        ops.mark("b:start")
        qd, qa = self.q1.head_dims, self.q1.head_acts
        ps = [self.q1.head(), self.q2.head(), self.q1.head(), self.q2.head(), self.tq1.head(), self.tq2.head()]
        ac = [self.qact["q1"], self.qact["q2"], self.qact_pi["q1"], self.qact_pi["q2"], self.qact_t["tq1"],
              self.qact_t["tq2"]]
        pb = [self.q1.head_bf16(), self.q2.head_bf16(), self.q1.head_bf16(), self.q2.head_bf16(), self.tq1.head_bf16(),
              self.tq2.head_bf16()]
        if self._gather_ok("q"):
            # Q inputs [enc(obs) | goal_enc | action] read in place: the state rows repeat every B rows (expand_obs on
            # embeddings), the actions are acts_main = [data | uniform | n x pi(obs) | n x pi(next)]; only the two
            # problems with weight gradients (q1 / q2 over all R rows) also write their assembled rows
            E = self.E
            seg = lambda k, mod, a_t: self._emb_segs(self._OBS_SRC[k][0], self._OBS_SRC[k][1] * B, self._OBS_SRC[k][2], mod) + [(a_t, 0, A, E, 0)]  # noqa: E731
            segs = [seg("q1", B, self.acts_main), seg("q2", B, self.acts_main), seg("q1", 0, self.act_pi), seg("q2", 0, self.act_pi),
                    seg("tq1", 0, self.act_next), seg("tq2", 0, self.act_next)]
            ops.mlp_fwd_gather(segs, [self.XQ["q1"], self.XQ["q2"], None, None, None, None], self.ldq, ps, pb, ac,
                               [self.R, self.R, B, B, B, B], qd, qa, lean=self._lean("q"))
        else:
            with ops.copy_batch():  # one launch
                for k in ("q1", "q2"):
                    ops.copy_cols(self.S[k], 0, self.lds, self.XQ[k], 0, self.ldq, self.R, self.E, src_row_mod=B)
                    ops.copy_cols(self.acts_main, 0, A, self.XQ[k], self.E, self.ldq, self.R, A)
                    ops.copy_cols(self.S[k], 0, self.lds, self.XQpi[k], 0, self.ldq, B, self.E)
                    ops.copy_cols(self.act_pi, 0, A, self.XQpi[k], self.E, self.ldq, B, A)
                for k in ("tq1", "tq2"):
                    ops.copy_cols(self.S[k], 0, self.lds, self.XT[k], 0, self.ldq, B, self.E)
                    ops.copy_cols(self.act_next, 0, A, self.XT[k], self.E, self.ldq, B, A)
            xs = [self.XQ["q1"], self.XQ["q2"], self.XQpi["q1"], self.XQpi["q2"], self.XT["tq1"], self.XT["tq2"]]
            ops.mlp_fwd(xs, self.ldq, ps, ac, [self.R, self.R, B, B, B, B], qd, qa, self.compute, params_bf16=pb, lean=self._lean("q"))
        qout = lambda buf, off, rows: buf[off: off + rows]  # noqa: E731
        q1m, q2m = qout(self.qact["q1"], self.q_yoff_R, self.R), qout(self.qact["q2"], self.q_yoff_R, self.R)
        q1p, q2p = qout(self.qact_pi["q1"], self.q_yoff_B, B), qout(self.qact_pi["q2"], self.q_yoff_B, B)
        t1, t2 = qout(self.qact_t["tq1"], self.q_yoff_B, B), qout(self.qact_t["tq2"], self.q_yoff_B, B)
        # Bellman + CQL (+Lagrange): losses and dL/dq for every row
        lap = self.log_alpha_prime
        call("tacorl_cql_loss", ptr(q1m), ptr(q2m), ptr(self.dq["q1"]), ptr(self.dq["q2"]), ptr(t1), ptr(t2),
             ptr(self.logp_cur), ptr(self.logp_nxt), ptr(self.logp_next), ptr(self.reward), ptr(self.done),
             ptr(self.log_alpha.param), ptr(lap.param) if self.with_lagrange else None, B, n, A,
             float(hp["discount"]), float(hp["reward_scale"]), float(hp["temp"]), float(hp["cons_w"]),
             float(hp["gap"]), int(hp["deterministic_backup"]), gs, ptr(lap.grad), ptr(self.logs), ptr(self.cql_ws),
             self.cql_ws.numel(), ops.stream())
        ops.mark("b:q_fwd+cql")
        # ---- actor backward: independent of the critic backward until the goal encoders, so it runs on a
        # side stream (a parallel branch of the captured graph); both are chains of small launches that
        # fill only part of the chip on their own
        if getattr(self, "_bwd_stream", None) is None:
            self._bwd_stream = torch.cuda.Stream(device=self.dev)
        main_stream = torch.cuda.current_stream()
        self._bwd_stream.wait_stream(main_stream)
        # (round 5: the slab reduces behind the five one-launch weight gradients below - readers: all-reduce and optimiser -
        # are recorded and leave as ONE launch at the end of the phase, ops.reduce_batch)
        with ops.reduce_batch(auto=self.R >= 16384):
            with torch.cuda.stream(self._bwd_stream):
                self._actor_backward(bc_phase, head_cur, q1p, q2p, gs)
                ops.mark("b:actor_bwd")
            # ---- critic backward through the Q MLPs; sum the broadcast embedding gradient over samples
            self._mlp_backward("q", [self.XQ["q1"], self.XQ["q2"]], self.ldq, [self.q1.head(), self.q2.head()],
                               [self.qact["q1"], self.qact["q2"]], [self.dq["q1"], self.dq["q2"]], 1,
                               [self.q1.head(self.q1.grad), self.q2.head(self.q2.grad)], [self.dXQ["q1"], self.dXQ["q2"]],
                               self.ldq, [self.R, self.R], qd, qa)
            call("tacorl_reduce_rows_mod_batch", 2, ops.ptr_array([self.dXQ["q1"], self.dXQ["q2"]]), self.ldq,
                 ops.ptr_array([self.dS["q1"], self.dS["q2"]]), self.lds, B, self.E, 3 * n + 1, ops.stream())
            ops.mark("b:critic_bwd")
            main_stream.wait_stream(self._bwd_stream)
            self._encoders_backward()
            ops.mark("b:enc_bwd")
            self._join_wgrads()
        ops.mark("b:end")

    lean_mlp_acts = True  # hidden-layer outputs of the fused MLPs are recomputed by the weight-gradient launch, not saved

    def _lean(self, tag):
        """True when the MLP site's forward may skip its hidden-layer outputs: its whole backward is the fused pair
        (input-gradient chain + one-launch weight gradients), the only reader of those outputs."""
        if self.compute != BF16 or not self.lean_mlp_acts:
            return False
        tag = {"qpi": "q"}.get(tag, tag)
        cache = self.__dict__.setdefault("_lean_cache", {})
        if tag not in cache:
            n, dims, ldx, ldo, ldd = {"genc": (5, self.actor.genc_dims, self.G, self.lds, self.G),
                                      "pi": (2, self.actor.head_dims, self.lds, self.HD, self.lds),
                                      "q": (6, self.q1.head_dims, self.ldq, 1, self.ldq)}[tag]
            cache[tag] = ops.mlp_lean_ok(n, dims, ldx, ldo, ldd, self.compute) and ops.mlp_bwd_fused_ok(n, dims, ldo, ldd, self.compute)
        return cache[tag]

    def _mlp_backward(self, tag, xs, ldx, params, acts_buf, d_outs, ldo, grads, d_xs, ldd, M, dims, acts):
        """MLP backward.  bf16 mode: the input-gradient chain is ONE launch on the current stream and the
        weight gradients go to a side stream of their own (a parallel graph branch joined before the
        optimiser) - they are needed only by Adam, not by anything downstream in the backward.
        Otherwise: the per-layer path."""
        n = len(xs)
        if not ops.mlp_bwd_fused_ok(n, dims, ldo, ldd, self.compute):
            ops.mlp_bwd(xs, ldx, params, acts_buf, d_outs, ldo, grads, d_xs, ldd, M, dims, acts, self.compute,
                        ws_tag="mlp_bwd_" + tag)
            return
        ops.mlp_bwd_fused_dgrad(params, acts_buf, d_outs, ldo, d_xs, ldd, M, dims, acts, "mlp_bwdf_" + tag,
                                prepacked=getattr(self, "_prepacked", False), lean=self._lean(tag))
        if all(g is None for g in grads):
            return
        side = self.wgrad_side_streams
        if not (side is True or (side and tag in side)):  # True / False / a collection of site tags
            ops.mlp_bwd_fused_wgrad(xs, ldx, acts_buf, d_outs, ldo, grads, M, dims, acts, "mlp_bwdf_" + tag, lean=self._lean(tag))
            return
        if not hasattr(self, "_wg_streams"):
            self._wg_streams, self._wg_pending = {}, []
        if tag not in self._wg_streams:
            self._wg_streams[tag] = torch.cuda.Stream(device=self.dev)
        ws_ = self._wg_streams[tag]
        ws_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws_):
            ops.mlp_bwd_fused_wgrad(xs, ldx, acts_buf, d_outs, ldo, grads, M, dims, acts, "mlp_bwdf_" + tag, lean=self._lean(tag))
        self._wg_pending.append(ws_)

    def _join_wgrads(self):
        for ws_ in getattr(self, "_wg_pending", []):
            torch.cuda.current_stream().wait_stream(ws_)
        self._wg_pending = []

    def _actor_backward(self, bc_phase, head_cur, q1p, q2p, gs):
        B, A, Ac, nz = self.B, self.A, self.Ac, self.noise
        qd, qa = self.q1.head_dims, self.q1.head_acts
        if bc_phase:
            call("tacorl_actor_head_bwd", ptr(head_cur), self.HD, ptr(nz["eps_pi"]), ptr(self.logp_pi), None, None, 0,
                 ptr(self.action), A, ptr(self.grip_pi) if self.dg else None, ptr(self.log_alpha.param), gs,
                 ptr(self.d_head), B, Ac, int(self.dg), ptr(self.logs), ops.stream())
        else:
            call("tacorl_actor_qmin", ptr(q1p), ptr(q2p), ptr(self.logp_pi), B, ptr(self.log_alpha.param),
                 ptr(self.dq_pi["q1"]), ptr(self.dq_pi["q2"]), gs, ptr(self.logs), ops.stream())
            self._mlp_backward("qpi", [self.XQpi["q1"], self.XQpi["q2"]], self.ldq, [self.q1.head(), self.q2.head()],
                               [self.qact_pi["q1"], self.qact_pi["q2"]], [self.dq_pi["q1"], self.dq_pi["q2"]], 1,
                               [None, None], [self.dXQpi["q1"], self.dXQpi["q2"]], self.ldq, [B, B], qd, qa)
            call("tacorl_actor_head_bwd", ptr(head_cur), self.HD, ptr(nz["eps_pi"]), ptr(self.logp_pi),
                 ops._at(self.dXQpi["q1"], self.E), ops._at(self.dXQpi["q2"], self.E), self.ldq, None, 0,
                 ptr(self.grip_pi) if self.dg else None, ptr(self.log_alpha.param), gs, ptr(self.d_head), B, Ac,
                 int(self.dg), ptr(self.logs), ops.stream())
        self._mlp_backward("pi", [self.S["a"]], self.lds, [self.actor.head()], [self.pact["a"]], [self.d_head], self.HD,
                           [self.actor.head(self.actor.grad)], [self.dS["a"]], self.lds, [B], self.actor.head_dims,
                           self.actor.head_acts)

    def _ebw_sequences(self):
        """Camera groups that share one conv-backward launch sequence: the cameras of one (fused-backward) geometry, at
        most 8 problems (3 networks per camera); everything else one camera at a time."""
        groups = {}
        for c in self.cams:
            groups.setdefault((tuple(self.hw[c]), bool(self._fused_bwd_ok(c))), []).append(c)
        merge = os.environ.get("TACORL_EBW_MERGE_CAMS", "1") == "1"
        seqs = []
        for (_, ok), cs in groups.items():
            if ok and merge and 3 * len(cs) <= 8:
                seqs.append(cs)
            else:
                seqs += [[c] for c in cs]
        return seqs

    def _encoders_backward(self):
        """Goal encoders (3 nets, one batch), then the three encoders (actor(obs, goal), q1, q2)."""
        B = self.B
        nets = {"a": self.actor, "q1": self.q1, "q2": self.q2}
        ks = ["a", "q1", "q2"]
        self._mlp_backward("genc", [self.gin[k] for k in ks], self.G, [nets[k].genc() for k in ks],
                           [self.gact[k] for k in ks], [ops._at(self.dS[k], self.Eo) for k in ks], self.lds,
                           [nets[k].genc(nets[k].grad) for k in ks], [self.dgin[k] for k in ks], self.G, [B] * 3,
                           self.actor.genc_dims, self.actor.genc_acts)
        ek = {"a": "a_og", "q1": "q1", "q2": "q2"}
        with ops.copy_batch():
            for j, c in enumerate(self.cams):
                for k in ks:
                    ops.copy_cols(self.dS[k], 32 * j, self.lds, self.enc_dout[(ek[k], c)], 0, 32, B, 32)
                    ops.copy_cols(self.dgin[k], 32 * j, self.G, self.enc_dout[(ek[k], c)], B * 32, 32, B, 32)
        # cameras of one geometry share a launch sequence (round 5; C4: both cameras 128 x 128 -> one 6-problem sequence
        # instead of two 3-problem ones - the conv-backward launches cost ~6-10 us each before their first image)
        for cs in self._ebw_sequences():
            c = cs[0]
            H, W = self.hw[c]
            imgs = [self._img_ptr(cc, 0) for cc in cs for _ in ks]
            ops_n = [2 * B] * (3 * len(cs))
            pa = [ops.ptr_array(imgs), ops.ptr_array([nets[k].enc(cc) for cc in cs for k in ks]),
                  ops.ptr_array([self.enc_act[(ek[k], cc)] for cc in cs for k in ks]),
                  ops.ptr_array([self.enc_dout[(ek[k], cc)] for cc in cs for k in ks]),
                  ops.ptr_array([nets[k].enc(cc, nets[k].grad) for cc in cs for k in ks]), ops.int_array(ops_n), H, W]
            np_ = 3 * len(cs)
            if self._fused_bwd_ok(c):  # per-image LDS-resident conv backward (encoder_bwd_fused.hip)
                nb = ops.L.lib().tacorl_encoder_bwd_fused_ws_bytes(np_, ops.int_array(ops_n), H, W)
                ws = ops.workspace(nb, self.dev, "enc_bwd_fused_" + "+".join(cs))
                pk = int(getattr(self, "_prepacked", False))
                img_p, par_p, act_p, dout_p, grad_p, n_p = pa[:6]
                # dependent chain: FC-tail input gradients (one launch) -> soft-argmax + conv backward
                call("tacorl_encoder_bwd_fused_head", np_, par_p, act_p, dout_p, n_p, H, W, pk, ptr(ws), ws.numel(), ops.stream())
                call("tacorl_encoder_bwd_fused_conv", np_, img_p, par_p, act_p, grad_p, n_p, H, W, 0, pk, ptr(ws), ws.numel(),
                     ops.stream())
                # FC weight gradients last and in line: on a side branch they ran beside the conv-backward
                # kernels, whose 255 one-per-CU workgroups then no longer fit in one round (+0.13 ms/step)
                call("tacorl_encoder_bwd_fused_fc_wgrad", np_, act_p, dout_p, grad_p, n_p, H, W, 0, ptr(ws), ws.numel(),
                     ops.stream())
                continue
            nb = ops.L.lib().tacorl_encoder_bwd_ws_bytes(3, ops.int_array(ops_n), H, W)
            ws = ops.workspace(nb, self.dev, "enc_bwd")
            call("tacorl_encoder_bwd", 3, ops.ptr_array(imgs), ops.ptr_array([nets[k].enc(c) for k in ks]),
                 ops.ptr_array([self.enc_act[(ek[k], c)] for k in ks]),
                 ops.ptr_array([self.enc_dout[(ek[k], c)] for k in ks]),
                 ops.ptr_array([nets[k].enc(c, nets[k].grad) for k in ks]), ops.int_array(ops_n), H, W,
                 BF16 if self.img_dtype == torch.bfloat16 else F32, self.compute, 0, ptr(ws), ws.numel(), ops.stream())

    def phase_c(self, optimize=True):
        """Optimiser steps (grads were all taken on the pre-step graph, as in the reference)."""
        hp, lap = self.hp, self.log_alpha_prime
        if optimize:
            a = self.actor
            items = [(lap.param, lap.grad, lap.m, lap.v, hp["critic_lr"], 0.0, lap.step, None, 0.0)] if self.with_lagrange else []
            mir = [(None, None)] * len(items)
            items.append((a.param, a.grad, a.m, a.v, hp["actor_lr"], hp["clip"], a.step, None, 0.0))
            mir.append((a.param_bf16, None))
            for q, t in ((self.q1, self.tq1), (self.q2, self.tq2)):
                items.append((q.param, q.grad, q.m, q.v, hp["critic_lr"], hp["clip"], q.step, t.param, hp["tau"]))
                mir.append((q.param_bf16, t.param_bf16))
            write = self.compute == BF16 and self.adam_writes_mirrors
            ops.adam_step_batch(items, mirrors=mir if write else None)  # two launches for all blocks
            if write:  # (this launch rewrote every mirror from the updated parameters, whatever state they were in)
                self.mirrors_written()
            if self.ef_pack_late:
                for c in self.cams:
                    if self._fused_ok(c):
                        self._pack_encoders(c, self._late_pack_nets())
        ops.mark("c:adam")

    def _allreduce(self, tensors):
        if D.collectives_on(self.world):
            for t in tensors:
                D.all_reduce_sum_(t)

    def metrics(self):
        """The logged scalars, read back (one D2H sync).  With several ranks they are first averaged over the ranks (one
        small collective on the steps that log): every slot is a per-rank batch mean or rank-independent, so the result is
        the full-batch value - what the reference's sync_dist=True logs give."""
        for fn in getattr(self, "pre_metrics", ()):  # device work only the logged scalars need (TACORL: the lazy action-decoder loss sum)
            fn()
        logs, div = D.reduce_logs_(self.logs, self.world)
        v = logs.cpu().tolist()
        return dict(zip(LOG_SLOTS, [x / div for x in v]))
