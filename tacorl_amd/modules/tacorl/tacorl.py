"""TACORL - drop-in for reference modules/tacorl/tacorl.py:21-300.

`training_step(batch)` = frozen LMP encoder over the B*T window frames -> plan-recognition
posterior -> sampled latent plan (the RL "action") -> (optional) action-decoder fine-tune ->
CQL update on (s, s', g) = (states[:,0], states[:,-1], goal), r = d = [disp == 1].
Select with `module._target_=tacorl_amd.modules.tacorl.tacorl.TACORL`.
"""
from pathlib import Path

import os

import torch
import torch.nn as nn

from ... import dist as D
from ... import ops
from ..._lib import BF16, F32, call, ptr
from ..common import register_views
from ..cql.cql_offline_lightning import CQL_Offline


class TACORL(CQL_Offline):
    def __init__(self, play_lmp_dir: str = "~/tacorl/models/play_lmp", lmp_epoch_to_load: int = -1,
                 overwrite_lmp_cfg: dict = {}, finetune_action_decoder: bool = False,
                 action_decoder_lr: float = 1e-4, *args, play_lmp=None, **kwargs):
        self.play_lmp_dir = Path(play_lmp_dir).expanduser()
        self.lmp_epoch_to_load = lmp_epoch_to_load
        self.overwrite_lmp_cfg = overwrite_lmp_cfg
        self.finetune_action_decoder = finetune_action_decoder
        self.action_decoder_lr = action_decoder_lr
        # The reference evaluates the action-decoder loss on every step even when the decoder is frozen,
        # although the value only reaches the logger (tacorl.py:206-233, optimize=False).  With
        # `action_loss_every_n_steps=k` that logging-only forward runs on every k-th step (match it to
        # Trainer(log_every_n_steps)); fine-tuning (`finetune_action_decoder=True`) always runs it.
        self.action_loss_every_n_steps = kwargs.pop("action_loss_every_n_steps", 1)
        # opt-in to a full unpickle of the PlayLMP checkpoint (real PL checkpoints carry DictConfig hyper-parameters)
        self.lmp_unsafe_pickle = bool(kwargs.pop("lmp_unsafe_pickle", False))
        self.__dict__["_play_lmp"] = play_lmp  # nn.Module: must not be registered as a sub-module
        super().__init__(*args, **kwargs)

    # ------------------------------------------------------------------ construction
    def build_networks(self):
        """reference tacorl.py:44-126."""
        from ..play_lmp.play_lmp_for_rl import PlayLMP, load_play_lmp

        lmp = self.__dict__.pop("_play_lmp", None)
        if lmp is None:
            lmp = load_play_lmp(self.play_lmp_dir, self.lmp_epoch_to_load, self.overwrite_lmp_cfg, device=self.dev,
                                compute_dtype=self.compute, image_dtype=self.img_dtype, unsafe_pickle=self.lmp_unsafe_pickle)
        assert isinstance(lmp, PlayLMP)
        from .. import cfgcheck

        cfgcheck.check_critic(self.critic_cfg, "critic")  # layers / hidden mirror the actor's (tacorl.py:72-73)
        cfgcheck.check_representation(self.critic_encoder_cfg, "critic_encoder", lmp.plan_proposal_obs_modalities)
        self.action_decoder_modalities = list(lmp.action_decoder_modalities)
        self.plan_recognition_modalities = list(lmp.plan_recognition_modalities)
        self.all_modalities = sorted(set(self.action_decoder_modalities + self.plan_recognition_modalities))
        cams, goal_cams = list(lmp.plan_proposal_obs_modalities), list(lmp.plan_proposal_goal_modalities)
        self.obs_modalities, self.goal_modalities = cams, goal_cams
        self.action_dim = lmp.pr.latent_plan_dim
        a = dict(policy_layers=lmp.policy_layers, q_layers=lmp.policy_layers, hidden=lmp.hidden,
                 discrete_gripper=False)  # critic q_network mirrors the actor's layers/hidden (tacorl.py:72-73)
        self._make_engine(cams, goal_cams, self.action_dim, a)
        e = self.engine
        # actor = deepcopy(LMP encoder) + deepcopy(LMP goal encoder) + LMP plan proposal (tacorl.py:63-70)
        e.actor.param.copy_(lmp.net.param)
        # critics: fresh encoders + fresh Q, goal encoders copied from the LMP (tacorl.py:93-120)
        from ...init import init_views_

        for q in (e.q1, e.q2):
            init_views_(q.views)
            n = q.head_off - q.genc_off
            q.param[q.genc_off: q.genc_off + n].copy_(lmp.net.param[lmp.net.genc_off: lmp.net.genc_off + n])
        self.sync_targets()
        self._register()
        # frozen LMP pieces (tacorl.py:51-53,124-126)
        self.lmp_net, self.pr, self.ad = lmp.net, lmp.pr, lmp.ad
        enc_views = {k[len("encoder."):]: v for k, v in lmp.net.views.items() if k.startswith("encoder.")}
        register_views(self, "perceptual_encoder.", enc_views, requires_grad=False)
        register_views(self, "plan_recognition.", lmp.pr.blk.views, requires_grad=False)
        if self.ad is not None:
            e.pre_metrics = [self.ad.finish_loss]  # (the logging-only decoder loss leaves its partial sums on the device: loss(lazy=True))
            self._pv["action_decoder"] = register_views(self, "action_decoder.", self.ad.blk.views)
            for k, v in self.ad.buffers.items():
                self.action_decoder.register_buffer(k, v)
            if self.finetune_action_decoder:
                # the decoder's gradients join [actor | q1 | q2 | log_alpha'] in the engine's arena: with several GPUs the
                # step's second all-reduce covers them - no third collective inside the first segment
                e.extend_arena(self.ad.blk)
        self._T = None
        # rollout surface (evaluation/rollout_manager.py:330-386): the frozen LMP encoder for the action decoder's state,
        # the decoder's one-step `act` with its carried hidden state
        from ..inference import attach_rollout_surface

        attach_rollout_surface(self, e.actor, cams, goal_cams, self.action_dim, False, lmp_net=self.lmp_net,
                               lmp_cams=self.action_decoder_modalities, ad=self.ad)

    def named_gradients(self):
        out = super().named_gradients()
        if self.ad is not None and self.finetune_action_decoder:
            out.update({f"action_decoder.{k}": v for k, v in self.ad.blk.grad_views.items()})
        return out

    # ---------------------------------------------------------------------- stepping
    def _ensure_seq(self, B, T, hw):
        if self._T == (B, T, tuple(sorted(hw.items()))):
            return
        ops.note_alloc()
        dev = self.dev
        self.frames = {c: torch.zeros(B * T, *hw[c], 3, device=dev, dtype=self.img_dtype) for c in self.all_modalities}
        self.f_out = {c: torch.zeros(B * T, 32, device=dev) for c in self.all_modalities}
        self.f_act = {c: torch.zeros(ops.encoder_act_layout(B * T, *hw[c])[1], device=dev) for c in self.all_modalities}
        npr = len(self.plan_recognition_modalities)
        self.pr_in = torch.zeros(B * T, 32 * npr, device=dev)
        # the sampled plan IS the RL action and the displacement flag IS reward and done (get_rl_batch,
        # tacorl.py:142-179): alias the engine's transition buffers instead of copying into them every step
        e = self.engine
        assert e.action.shape == (B, self.action_dim) and e.action.dtype == torch.float32
        self.plan, self.reward = e.action, e.reward
        # the frozen LMP encoder over the B*T window frames rides in the engine's encoder launches
        self.engine.extra_enc = [dict(cam=c, img=self.frames[c], net=self.lmp_net, out=self.f_out[c],
                                      act=self.f_act[c], n=B * T) for c in self.all_modalities]
        self._T = (B, T, tuple(sorted(hw.items())))

    def _stage_small(self, batch, noise, B, T):
        e = self.engine
        disp, acts = batch["disp"], batch["actions"] if self.ad is not None else None
        dd = {torch.float32: 0, torch.int64: 1, torch.int32: 2, torch.uint8: 3, torch.bool: 3}.get(disp.dtype)
        if (dd is not None and disp.is_cuda and disp.is_contiguous() and disp.numel() == B
                and (acts is None or (acts.is_cuda and acts.is_contiguous() and acts.dtype == torch.float32
                                      and acts.shape == self.acts.shape))):
            # reward = done = (disp == 1) and the action window, one launch
            call("tacorl_stage_transition", ptr(disp), dd, ptr(self.reward), ptr(e.done), B,
                 ptr(acts) if acts is not None else None, ptr(self.acts) if acts is not None else None,
                 acts.numel() if acts is not None else 0, ops.stream())
        else:
            if acts is not None:
                self.acts.copy_(acts)
            self.reward.copy_(disp == 1)
            e.done.copy_(self.reward)
        e.set_noise(noise)

    def _stage_frames(self, batch, noise, nchw=True):
        """Eager part of the step: pack the window frames (reference NCHW fp32 -> NHWC image dtype) into
        fixed buffers, copy the small tensors, draw / copy the noise.
        uint8 frames - the dataset's own format, (B, T, H, W, 3) with goal (B, H, W, 3) - are taken as they are
        and normalised on the way (ToTensor + Normalize(0.5, 0.5), bit-identical to the transformed fp32 frames)."""
        from ...data.replay import wait_ready

        wait_ready(batch)  # a replay batch's small tables travel on the feeder's copy stream
        rp = batch.get("replay")  # frames by index out of a uint8 dataset (data/replay.py HbmReplay.batch(fused=True))
        if rp is not None:
            u8, nchw, B, T = True, False, rp["B"], rp["T"]
            states = rp["frames"]
            hw = {c: tuple(v.shape[1:3]) for c, v in states.items()}
        else:
            states = batch["states"]
            u8 = next(iter(states.values())).dtype == torch.uint8
            if u8:
                nchw = False
            B, T = next(iter(states.values())).shape[:2]
            hw = {c: (tuple(v.shape[-2:]) if nchw else tuple(v.shape[-3:-1])) for c, v in states.items()}
        src_hw = dict(hw)  # the frames as stored; an augmentation spec with a Resize stage sets the encoders' geometry
        rs = (batch.get("aug") or {}).get("resize") or {}
        if rs:
            if not u8:
                raise ValueError("aug['resize'] needs the dataset's uint8 frames (the resize is part of the uint8 pack)")
            hw = {c: tuple(rs.get(c, hw[c])) for c in hw}
        self.engine.extra_normal = {"eps_pr": (B, self.action_dim)}  # drawn with the engine's noise (one launch)
        self.engine.ensure_batch(B, {c: hw[c] for c in self.engine.cams})
        self.eps_pr = self.engine.extra_noise["eps_pr"]
        self._ensure_seq(B, T, hw)
        xd = BF16 if self.img_dtype == torch.bfloat16 else F32
        e = self.engine
        if self.ad is not None and (getattr(self, "acts", None) is None or self.acts.shape[:2] != (B, T)):
            ops.note_alloc()
            self.acts = torch.zeros(B, T, 7, device=self.dev)
        # the small launches of the eager part (reward / done / action window, the two noise generators) go beside the
        # HBM-bound image pack on a second stream instead of behind it
        if getattr(self, "_stage_stream", None) is None:
            self._stage_stream = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream()
        side = self._stage_stream if getattr(self, "stage_side", True) else main
        side.wait_stream(main)  # the previous step's graph read these buffers
        with torch.cuda.stream(side):
            self._stage_small(batch, noise, B, T)
        # get_rl_batch (tacorl.py:142-179) as strided views: s = states[:,0], s' = states[:,-1]
        for c in sorted(set(self.all_modalities) | set(e.cams)):
            H, W = hw[c]
            v = states[c]
            sz = 1 if u8 else 4  # bytes per source element
            assert v.is_cuda and v.is_contiguous() and v.dtype == (torch.uint8 if u8 else torch.float32)
            if rp is not None:
                # image i of a job = dataset frame ids[i * stride]: the window (stride 1), obs = ids[b T], goal = the table's
                # tail, next = ids[b T + T - 1] - gather and pack in one pass over the dataset
                Hs, Ws = src_hw[c]
                ids, fb = rp["ids"], 3 * Hs * Ws
                assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == B * T + B
                ip = ids.data_ptr()
                jobs = [(v.data_ptr(), fb, self.frames[c].data_ptr(), B * T, ip, 1)] if c in self.all_modalities else []
                if c in e.cams:
                    esz, x3, ob = e.X3[c].element_size(), e.X3[c].data_ptr(), 3 * H * W  # (ob: elements per packed image)
                    jobs += [(v.data_ptr(), fb, x3, B, ip, T), (v.data_ptr(), fb, x3 + B * ob * esz, B, ip + 8 * B * T, 1),
                             (v.data_ptr(), fb, x3 + 2 * B * ob * esz, B, ip + 8 * (T - 1), T)]
                if fb % 16 or v.data_ptr() % 16:
                    raise ValueError("uint8 dataset: H*W*3 must be a multiple of 16 and the tensor 16-byte aligned")
                aug = batch.get("aug")
                if aug is None:
                    ops.pack_images_u8_gather_batch(jobs, xd, H, W)
                else:
                    st, gl = aug["states"][c], aug["goal"][c]
                    sh, ji = st.get("shift"), st.get("jitter")
                    row = lambda t, k: None if t is None else t[:, k].contiguous()  # noqa: E731
                    flat = lambda t: None if t is None else t.reshape(B * T, t.shape[-1]).contiguous()  # noqa: E731
                    tabs = [(flat(sh), flat(ji))] if c in self.all_modalities else []
                    if c in e.cams:
                        tabs += [(row(sh, 0), row(ji, 0)), (gl.get("shift"), gl.get("jitter")), (row(sh, T - 1), row(ji, T - 1))]
                    ops.pack_images_u8_resize_aug_batch([j + t for j, t in zip(jobs, tabs)], xd, (Hs, Ws), H, W, aug["pad"][c])
                continue
            Hs, Ws = src_hw[c]
            simg = 3 * Hs * Ws  # elements of a source frame
            jobs = [(v.data_ptr(), simg, self.frames[c].data_ptr(), B * T)] if c in self.all_modalities else []
            if c in e.cams:
                g = batch["goal"][c]
                assert g.is_cuda and g.is_contiguous() and g.dtype == v.dtype
                esz, img = e.X3[c].element_size(), H * W * 3
                x3 = e.X3[c].data_ptr()
                jobs += [(v.data_ptr(), T * simg, x3, B), (g.data_ptr(), simg, x3 + B * img * esz, B),
                         (v.data_ptr() + sz * (T - 1) * simg, T * simg, x3 + 2 * B * img * esz, B)]
            aug = batch.get("aug") if u8 else None
            if aug is not None:
                # train-time augmentations on the way in: the draws arrive as device tables; the obs / next
                # frames of the transition take the draws of window frames 0 / T-1, as in the reference, where the
                # transform ran once per window in the dataset
                st, gl = aug["states"][c], aug["goal"][c]
                sh, ji = st.get("shift"), st.get("jitter")
                row = lambda t, k: None if t is None else t[:, k].contiguous()  # noqa: E731
                flat = lambda t: None if t is None else t.reshape(B * T, t.shape[-1]).contiguous()  # noqa: E731
                tabs = [(flat(sh), flat(ji))] if c in self.all_modalities else []
                if c in e.cams:
                    tabs += [(row(sh, 0), row(ji, 0)), (gl.get("shift"), gl.get("jitter")), (row(sh, T - 1), row(ji, T - 1))]
                ops.pack_images_u8_resize_aug_batch([j + (None, 1) + t for j, t in zip(jobs, tabs)], xd, (Hs, Ws), H, W,
                                                    aug["pad"][c])
            elif u8:
                if (H * W * 3) % 16 or any(j[0] % 16 for j in jobs):
                    raise ValueError("uint8 frames: H*W*3 must be a multiple of 16 and the tensors 16-byte aligned")
                ops.pack_images_u8_batch(jobs, xd, H, W)  # pitches are in bytes = elements
            elif nchw and (H * W) % 4 == 0:  # one launch for the window frames and the obs / goal / next images
                import ctypes as C
                if (c in self.all_modalities and c in e.cams and len(jobs) == 4 and T >= 2
                        and os.environ.get("TACORL_PACK_DEDUP", "1") == "1"):
                    # obs = window frame 0 and next = window frame T - 1 (get_rl_batch): written from the one read of the
                    # window; the goal image stays a job of its own
                    win, _, goal, _ = jobs
                    x3, img_b = e.X3[c].data_ptr(), H * W * 3 * e.X3[c].element_size()
                    js = [win, goal]
                    call("tacorl_pack_images_window_batch", 2, (C.c_void_p * 2)(*[j[0] for j in js]),
                         (C.c_long * 2)(*[j[1] for j in js]), (C.c_void_p * 2)(*[j[2] for j in js]),
                         ops.int_array([j[3] for j in js]), (C.c_void_p * 2)(x3, None), (C.c_void_p * 2)(x3 + 2 * B * img_b, None),
                         ops.int_array([T, 0]), xd, H, W, ops.stream())
                    continue
                call("tacorl_pack_images_batch", len(jobs), (C.c_void_p * len(jobs))(*[j[0] for j in jobs]),
                     (C.c_long * len(jobs))(*[j[1] for j in jobs]), (C.c_void_p * len(jobs))(*[j[2] for j in jobs]),
                     ops.int_array([j[3] for j in jobs]), xd, H, W, ops.stream())
            else:
                for src, pitch, dst, n in jobs:
                    call("tacorl_pack_images", src, pitch, int(nchw), dst, xd, n, 3, H, W, ops.stream())
        main.wait_stream(side)
        return B, T, hw

    def _ad_due(self):
        k = self.action_loss_every_n_steps
        return self.ad is not None and (self.finetune_action_decoder or k <= 1 or (self._step_count + 1) % k == 0)

    def _device_front(self, B, T, hw, optimize, with_ad=True):
        """Graph-capturable: all encoders (frozen LMP + actor/critics/targets) -> plan recognition -> plan ->
        AD loss -> first phase of the CQL update (up to the alpha gradient)."""
        e = self.engine
        ops.mark("front:start")
        if getattr(self, "_pr_stream", None) is None:
            # (TACORL_PR_PRIO=1, with the encoder-split experiment: the plan recognition's stream at high priority, so that
            # its workgroups take the CUs the first encoder launch frees ahead of the second launch's)
            prio = -1 if os.environ.get("TACORL_PR_PRIO", "0") == "1" else 0
            self._pr_stream, self._side_stream = torch.cuda.Stream(device=self.dev, priority=prio), torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream()
        mods = self.plan_recognition_modalities
        # one camera: the frame embeddings are the transformer's input as they stand
        emb, ld = (self.f_out[mods[0]], 32) if len(mods) == 1 else (self.pr_in, self.pr_in.shape[1])
        # fine-tuning: the decoder's weights-only preparation (bf16 mirrors, transposed copies for BPTT and the heads' input
        # gradient: ~45 us at the head of the decoder's chain, which is the step's critical path in C3) beside the encoders
        ad_prep, ad_prepared = None, False
        if (with_ad and optimize and self.finetune_action_decoder and getattr(self, "ad_early_prepare", True)
                and self.ad._bwd_fast(B, self.compute)):
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                self.ad.refresh_mirrors(B, T - 1)
                ad_prepared = self.ad.prepare_backward(B, T - 1, self.compute)
                ad_prep = torch.cuda.Event()
                ad_prep.record(self._side_stream)
        # The frozen LMP window's encoder problems as a launch of their own FIRST, the plan recognition -> action decoder
        # branch forked right behind it, the update's own encoder problems after that on 160 workgroups (engine.encode_split):
        # where that branch is the step's longer chain it starts ~80 us earlier.  Measured (round 5): C4's share (window 32: 33
        # recurrent launches of 64 rows; `ad:end` 1 017 us against `c:adam` 864) 1.274 - 1.293 -> 1.222 - 1.238 ms/step in six
        # of seven runs (one read 1.337: whose workgroups get the freed CUs first is a race); the headline step (window 16,
        # balanced chains) +12 us; the same step with the decoder fine-tuned (C3: loss, BPTT, weight gradients and Adam make
        # that branch the critical one) 1.518 -> 1.497.  TACORL_EF_SPLIT_LMP = auto (default: windows of 24 steps and more, or
        # a fine-tuned decoder) / 0 / 1.
        sp = os.environ.get("TACORL_EF_SPLIT_LMP", "auto")
        split = with_ad and (sp == "1" or (sp == "auto" and (T >= 24 or (optimize and self.finetune_action_decoder))))
        forked = []

        def fork_branches():
            forked.append(self._fork_pr_ad(B, T, main, mods, emb, ld, with_ad, optimize, ad_prep, ad_prepared))

        if not (split and e.encode_split(fork_branches)):
            e._encode_all()
        ops.mark("front:encoded")
        if not forked:
            fork_branches()
        ready, ad_on_side = forked[0]
        e.action_ready = ready
        e.phase_a(encoded=True, optimize=optimize)
        if ad_prep is not None:
            main.wait_stream(self._side_stream)  # (every forked stream joins the capture's origin)
        segmented = self._segmented() or not self._use_graph
        if with_ad and not ad_on_side:
            if segmented:
                main.wait_stream(self._pr_stream)
            else:
                # one graph for the whole step: the fine-tuning chain (loss, BPTT, weight gradients, Adam - 1.4 ms, touching
                # nothing but the decoder's own buffers) stays a branch beside the CQL update and joins at the end of the step -
                # or, with the collectives captured inside that graph (several ranks), in front of the SECOND all-reduce, whose
                # arena holds the decoder's gradients (round 6: it used to join here, in front of the first one - the whole
                # decoder chain then stood between phase_a and phase_b: 1.078 against 0.864 ms at C3's B = 32 share)
                self._ad_join = self._pr_stream
        if segmented:
            # every segment must be self-contained (its graph is replayed on its own)
            main.wait_stream(self._pr_stream)
            e.action_ready = None
            self._join_ad()

    def _fork_pr_ad(self, B, T, main, mods, emb, ld, with_ad, optimize, ad_prep, ad_prepared):
        """The plan recognition -> plan -> action-decoder branches of the step, forked from `main` where it stands.  Returns
        (the event "plan / RL action written", whether the decoder pass runs on the second side stream)."""
        e = self.engine
        # Plan recognition -> plan -> (action-decoder loss) only need the frame embeddings; the first phase of
        # the CQL update does not need the plan.  They run as parallel branches of the step's graph:
        #   side stream 1: PR transformer -> sampled plan (= the RL action, in place) -> event action_ready
        #   side stream 2 (forked after the plan): frozen action-decoder forward + loss (logging only)
        #   main stream  : policy / sampling / alpha (phase_a); phase_b waits for action_ready
        self._ad_join = None
        ad_on_side = with_ad and not (optimize and self.finetune_action_decoder)
        self._pr_stream.wait_stream(main)
        with torch.cuda.stream(self._pr_stream):
            ops.mark("pr:start")
            if len(mods) > 1:
                for j, c in enumerate(mods):
                    ops.copy_cols(self.f_out[c], 0, 32, self.pr_in, 32 * j, self.pr_in.shape[1], B * T, 32)
            # the LMP networks are frozen in TACORL (tacorl.py:52-66): their weight-only preparation is cached
            self.pr.forward(emb, ld, B, T, self.compute, inference=True, sample=(self.eps_pr, self.plan), frozen=True)
            ops.mark("pr:plan")
            ready = torch.cuda.Event()
            ready.record(self._pr_stream)
            if with_ad and not ad_on_side:
                # fine-tuning: loss + BPTT (+ Adam on one GPU) stay in line on this branch.  With more than one GPU the
                # decoder's gradient block is part of the engine's arena: the step's second all-reduce covers it and
                # the Adam step follows in the last segment (reference tacorl.py:206-233 has no dependency
                # between this update and the CQL update)
                if ad_prep is not None:
                    self._pr_stream.wait_event(ad_prep)
                self.ad.loss_step(self, self.acts, self.plan, B, T, True, defer_update=self._defer_ad_update(),
                                  mirrors_current=ad_prep is not None, prepared=ad_prepared)
        if ad_on_side:
            # compute_action_decoder_update (tacorl.py:206-233), frozen decoder: 30 small dependent GEMMs that each
            # fill a fraction of the chip -> their own branch, joined at the end of the step
            self._side_stream.wait_event(ready)
            with torch.cuda.stream(self._side_stream):
                ops.mark("ad:start")
                self.ad.loss_step(self, self.acts, self.plan, B, T, False, frozen=not self.finetune_action_decoder)
                ops.mark("ad:end")
            self._ad_join = self._side_stream
        return ready, ad_on_side

    def _defer_ad_update(self):
        """More than one rank (or the split-graph test mode): the fine-tuned decoder's Adam step waits for the arena's
        all-reduce.  On one GPU it stays on the decoder's own branch of the step's graph."""
        return D.collectives_on(self.world_size) or getattr(self, "_force_graph_split", False)

    def _join_ad(self):
        if getattr(self, "_ad_join", None) is not None:
            torch.cuda.current_stream().wait_stream(self._ad_join)
            self._ad_join = None

    def get_pr_latent_plan(self, batch, noise=None, nchw=True):
        """reference tacorl.py:235-252 (no_grad / eval): returns the sampled latent plan (device tensor)."""
        B, T, hw = self._stage_frames(batch, noise, nchw)
        xd = BF16 if self.img_dtype == torch.bfloat16 else F32
        for c in self.all_modalities:
            H, W = hw[c]
            call("tacorl_encoder_fwd", 1, ops.ptr_array([self.frames[c]]), ops.ptr_array([self.lmp_net.enc(c)]),
                 ops.ptr_array([self.f_out[c]]), ops.ptr_array([self.f_act[c]]), ops.int_array([B * T]), H, W, xd,
                 self.compute, ops.stream())
        for j, c in enumerate(self.plan_recognition_modalities):
            ops.copy_cols(self.f_out[c], 0, 32, self.pr_in, 32 * j, self.pr_in.shape[1], B * T, 32)
        head = self.pr.forward(self.pr_in, self.pr_in.shape[1], B, T, self.compute, inference=True)
        call("tacorl_pr_sample", ptr(head), ptr(self.eps_pr), ptr(self.plan), None, None, B, self.action_dim,
             float(self.pr.min_std), ops.stream())
        return self.plan

    def training_step(self, batch, batch_idx=0, noise=None):
        self._step(batch, noise, optimize=True, log_type="train")
        self._tick_optimizers()

    def validation_step(self, batch, *args, noise=None, **kwargs):
        self._step(batch, noise, optimize=False, log_type="validation")

    def _step(self, batch, noise, optimize, log_type, nchw=True):
        B, T, hw = self._stage_frames(batch, noise, nchw)
        with_ad = self._ad_due()
        key = ("tacorl", B, T, tuple(sorted(hw.items())), self.current_epoch < self.bc_epochs, optimize, with_ad)
        e, bc = self.engine, self.current_epoch < self.bc_epochs
        ad_update = with_ad and optimize and self.finetune_action_decoder and self._defer_ad_update()

        def tail():
            e.phase_c(optimize)
            self._join_ad()
            if ad_update:
                self.ad.update(self)  # its gradients came through the second all-reduce with the arena
            ops.mark("step:end")

        # split (multi-GPU) graphs: the frozen, logging-only action-decoder pass leaves the first segment
        # and runs as a side graph beside the all-reduces and the other segments
        segmented = self._segmented()
        ad_ft = optimize and self.finetune_action_decoder
        ad_side = with_ad and segmented and self._use_graph
        # (round 6: the FINE-TUNED decoder's loss + backward as the side graph too - joined in front of all-reduce #2, whose arena
        # holds its gradients; inside the first segment its chain, the step's longest, stood in front of phase_b)
        side = None
        if ad_side and ad_ft:
            side = (0, lambda: self.ad.loss_step(self, self.acts, self.plan, B, T, True, defer_update=True), 1)
        elif ad_side:
            side = (0, lambda: self.ad.loss_step(self, self.acts, self.plan, B, T, False, frozen=not self.finetune_action_decoder))
        def allreduce_grads():
            if ad_update:  # a fine-tuned decoder's branch: its gradient block is part of the arena (the frozen, logging-only
                self._join_ad()  # pass keeps running beside the all-reduce and the optimiser and joins at the end of the step)
            e.allreduce_grads()

        self._run_segments(key, [lambda: self._device_front(B, T, hw, optimize, with_ad and not ad_side),
                                 lambda: e.phase_b(bc, optimize), tail],
                           [e.allreduce_alpha, allreduce_grads], side=side)
        self._publish_logs(log_type, extra=("action_loss",) if with_ad else ())

    def configure_optimizers(self):
        o = super().configure_optimizers()
        if self.finetune_action_decoder and self.ad is not None:  # reference tacorl.py:289-300
            blk = self.ad.blk
            o.append(self._make_adam("action_decoder", [(blk, self._pv["action_decoder"], blk.views_of(blk.m),
                                                         blk.views_of(blk.v))], self.action_decoder_lr))
        return o
