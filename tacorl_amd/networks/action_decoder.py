"""ActionDecoderLogistic on HIP kernels (reference
networks/action_decoders/action_decoder_logistic.py:21-300, rnn_models.py:5-16):
2-layer ReLU RNN (hidden 2048) over [plan | perceptual emb] -> mixture heads -> discretised
logistic-mixture NLL + gripper cross-entropy.

Internals are time-major ([t][b] rows) so every per-step slice and every BPTT operand pair is a
contiguous row range; the four heads (mean_fc | log_scale_fc | prob_fc | gripper_fc) are stored
back to back and evaluated as one GEMM.
"""
import ctypes as C
import os

import torch

from .. import ops
from .._lib import ACT_NONE, ACT_RELU, call, ptr
from ..blocks import TensorBlock


class ActionDecoderLogistic:
    def __init__(self, device, state_dim=32, goal_dim=32, latent_plan_dim=16, hidden_size=256, out_features=7,
                 act_max_bound=(1.0,) * 7, act_min_bound=(-1.0,) * 7, gripper_alpha=1.0, policy_rnn_dropout_p=0.0,
                 num_layers=2, rnn_model="rnn_decoder", discrete_gripper=True, include_goal=False, num_classes=10,
                 n_mixtures=10, **unused):
        if rnn_model != "rnn_decoder" or not discrete_gripper or include_goal or policy_rnn_dropout_p != 0.0:
            raise NotImplementedError("only the configured decoder (relu nn.RNN, discrete gripper, no goal) is in scope")
        if any(abs(b - 1.0) > 0 for b in act_max_bound) or any(abs(b + 1.0) > 0 for b in act_min_bound):
            raise NotImplementedError("action bounds other than +-1 (config/networks/action_decoder/logistic.yaml)")
        self.dev, self.P, self.E, self.hidden, self.L = device, latent_plan_dim, state_dim, hidden_size, num_layers
        self.Da, self.K, self.num_classes, self.gripper_alpha = out_features - 1, n_mixtures, num_classes, gripper_alpha
        self.include_goal = include_goal
        H, In, DK = hidden_size, state_dim + latent_plan_dim, (out_features - 1) * n_mixtures
        spec = []
        for l in range(num_layers):
            spec += [(f"rnn.weight_ih_l{l}", (H, In if l == 0 else H)), (f"rnn.weight_hh_l{l}", (H, H)),
                     (f"rnn.bias_ih_l{l}", (H,)), (f"rnn.bias_hh_l{l}", (H,))]
        spec += [("mean_fc.weight", (DK, H)), ("log_scale_fc.weight", (DK, H)), ("prob_fc.weight", (DK, H)),
                 ("gripper_fc.weight", (2, H)), ("mean_fc.bias", (DK,)), ("log_scale_fc.bias", (DK,)),
                 ("prob_fc.bias", (DK,)), ("gripper_fc.bias", (2,))]
        self.blk = TensorBlock(spec, device)
        self.NH = 3 * DK + 2
        assert DK % 4 == 0, "head packing needs Da*K % 4 == 0"
        t = lambda v: torch.tensor(v, device=device, dtype=torch.float32)  # noqa: E731
        mx, mn = t(list(act_max_bound)[:-1]), t(list(act_min_bound)[:-1])
        self.buffers = {  # register_buffer'ed by the reference (state-dict interchange)
            "one_hot_embedding_eye": torch.eye(n_mixtures, device=device),
            "ones": torch.ones(1, 1, n_mixtures, device=device),
            "gripper_bounds": t([act_min_bound[-1], act_max_bound[-1]]),
            "action_max_bound": mx.view(1, 1, -1, 1) * torch.ones(1, 1, 1, n_mixtures, device=device),
            "action_min_bound": mn.view(1, 1, -1, 1) * torch.ones(1, 1, 1, n_mixtures, device=device),
        }
        self._shape = None
        self.hidden_state = None

    def clear_hidden_state(self):
        """reference :73-74."""
        self.hidden_state = None

    def act(self, latent_plan, perceptual_emb, latent_goal=None, noise=None, compute=None):
        """ActionDecoderLogistic.act (reference :87-97): ONE decoder step per call with the RNN's hidden state carried
        between calls (rollout: evaluation/rollout_manager.py:375-386).  latent_plan (B,P); perceptual_emb (B,1,E);
        returns the sampled action (B,1,7).  noise = (rand_a (B,1,6,K), rand_b (B,1,6)) injects the two U(0,1) draws
        of `_sample` (:247,258)."""
        if latent_goal is not None:
            raise NotImplementedError("include_goal=False is the configured decoder")
        compute = ops.F32 if compute is None else compute
        B, H, L = latent_plan.shape[0], self.hidden, self.L
        assert perceptual_emb.shape[0] == B and perceptual_emb.shape[1] == 1, "act decodes one step per call"
        f = lambda *s: torch.zeros(*s, device=self.dev)  # noqa: E731
        if getattr(self, "_act_B", None) != B:
            ops.note_alloc()
            self._act_x, self._act_xin = f(B, self.P + self.E), f(B, H)
            self._act_h = [[f(B, H) for _ in range(L)] for _ in range(2)]  # ping-pong: previous / new hidden state
            self._act_heads, self._act_out = f(B, (self.NH + 31) // 32 * 32), f(B, self.Da + 1)
            self._act_ra, self._act_rb = f(B, self.Da, self.K), f(B, self.Da)
            nb = ops.L.lib().tacorl_linear_add_fwd_ws_bytes(1, ops.int_array([B]), H, H)
            self._act_ws = torch.empty(max(256, nb), dtype=torch.uint8, device=self.dev)
            self._act_B, self._act_flip = B, 0
            self.hidden_state = None
        blk = self.blk
        plan = latent_plan.to(self.dev, torch.float32).contiguous()
        emb = perceptual_emb.to(self.dev, torch.float32).reshape(B, self.E).contiguous()
        call("tacorl_build_ad_input", ptr(plan), ptr(emb), self.E, ptr(self._act_x), B, 1, 1, self.P, self.E, ops.stream())
        prev = self._act_h[self._act_flip]
        new = self._act_h[self._act_flip ^ 1]
        if self.hidden_state is None:  # h_0 = 0 (nn.RNN default)
            for t in prev:
                t.zero_()
        x, K = self._act_x, self.P + self.E
        for l in range(L):
            self._lin(x, K, blk.p(f"rnn.weight_ih_l{l}"), blk.p(f"rnn.bias_ih_l{l}"), self._act_xin, B, K, H, ACT_NONE, compute)
            call("tacorl_linear_add_fwd", 1, ops.ptr_array([prev[l]]), H, ops.ptr_array([blk.p(f"rnn.weight_hh_l{l}")]),
                 ops.ptr_array([blk.p(f"rnn.bias_hh_l{l}")]), ops.ptr_array([self._act_xin]), H, ops.ptr_array([new[l]]), H,
                 ops.int_array([B]), H, H, ACT_RELU, compute, ptr(self._act_ws), self._act_ws.numel(), ops.stream())
            x, K = new[l], H
        self._lin(x, H, blk.p("mean_fc.weight"), blk.p("mean_fc.bias"), self._act_heads, B, H, self.NH, ACT_NONE, compute,
                  ldy=self._act_heads.shape[1])
        if noise is not None:
            self._act_ra.copy_(noise[0].reshape(self._act_ra.shape))
            self._act_rb.copy_(noise[1].reshape(self._act_rb.shape))
        else:
            self._act_ra.uniform_()
            self._act_rb.uniform_()
        call("tacorl_logistic_mixture_sample", ptr(self._act_heads), self._act_heads.shape[1], ptr(self._act_ra),
             ptr(self._act_rb), ptr(self._act_out), B, self.Da, self.K, ops.stream())
        self._act_flip ^= 1
        self.hidden_state = torch.stack(new)  # (num_layers, B, H) as nn.RNN's h_n
        return self._act_out.view(B, 1, self.Da + 1).clone()

    def _ensure(self, B, Tm):
        if self._shape == (B, Tm):
            return
        ops.note_alloc()
        f = lambda *s: torch.zeros(*s, device=self.dev)  # noqa: E731
        R, H = B * Tm, self.hidden
        self.x_seq = f(R, self.P + self.E)
        self.xin = [f(R, H) for _ in range(self.L)]
        self.h = [f(R, H) for _ in range(self.L)]
        self.h0 = f(B, H)
        self.NHP = (self.NH + 31) // 32 * 32  # row pitch of the head buffers (ring GEMM: N % 32 == 0)
        self.heads, self.d_heads = f(R, self.NHP), f(R, self.NHP)
        self.ws = torch.empty(max(256, ops.L.lib().tacorl_logistic_mixture_ws_bytes(B, Tm, self.Da)), dtype=torch.uint8,
                              device=self.dev)
        nb = ops.L.lib().tacorl_linear_add_fwd_ws_bytes(1, ops.int_array([B]), H, H)
        self.rnn_ws = torch.empty(max(256, nb), dtype=torch.uint8, device=self.dev)
        # bf16 mode: hidden states and the H x H recurrent weights also live as bf16 (the ring GEMM's operands)
        bf = lambda *s: torch.zeros(*s, device=self.dev, dtype=torch.bfloat16)  # noqa: E731
        # (rows padded to a multiple of 64 with zeros: the top layer's copy is also the row-slabbed heads weight gradient's operand)
        self.hb = [bf((R + 63) // 64 * 64, H) for _ in range(self.L)]
        self.h0b = bf(B, H)
        self.whb = [bf(H, H) for _ in range(self.L)]
        self.wib = [None] + [bf(H, H) for _ in range(1, self.L)]  # W_ih of layers >= 1 (H x H)
        self.headw_b, self.headb = bf(self.NHP, H), f(self.NHP)    # output heads, rows padded with zeros
        # layer 0's input projection inside the ring GEMM (K extension, proj_in_ring): the bf16 input rows, zero padded to 128
        # columns, and W_ih_l0 likewise
        self.xb_seq, self.wih0_b = bf(R, 128), bf(H, 128)
        self._shape = (B, Tm)

    # ------------------------------------------------------------------ twin pass (logging-only second plan)
    def twin_ok(self, B, compute):
        """A second, logging-only pass (other plan, same frames, same weights) can ride in the launches of forward():
        the ring-GEMM path with the fused input projection (bf16)."""
        H, K = self.hidden, self.P + self.E
        return (compute == ops.BF16 and bool(ops.L.lib().tacorl_rnn_linear_supported(B, H, H)) and 32 <= K <= 64
                and K % 8 == 0 and H % 16 == 0 and getattr(self, "fused_input_proj", True) and getattr(self, "twin_pass", True))

    def twin_state(self, B, Tm):
        """Activation buffers of the twin pass (the weights and their bf16 mirrors are this decoder's)."""
        tw = getattr(self, "_twin", None)
        if tw is None or tw.shape != (B, Tm):
            ops.note_alloc()
            self._ensure(B, Tm)
            R, H = B * Tm, self.hidden
            f = lambda *s: torch.zeros(*s, device=self.dev)  # noqa: E731
            bf = lambda *s: torch.zeros(*s, device=self.dev, dtype=torch.bfloat16)  # noqa: E731
            tw = self._twin = type("TwinPass", (), {})()
            tw.shape = (B, Tm)
            tw.xin, tw.h, tw.hb = [f(R, H) for _ in range(self.L)], [f(R, H) for _ in range(self.L)], [bf(R, H) for _ in range(self.L)]
            tw.heads = f(R, self.NHP)
            tw.xb_seq = bf(R, 128)
            tw.ws = torch.empty(self.ws.numel(), dtype=torch.uint8, device=self.dev)
        return tw

    # Round 5: layer 0's input projection W_ih [plan | emb_t] + b_ih is no launch of its own (tacorl_ad_input_proj: 19.8 us at the
    # head of the action-decoder branch, 33.5 MB of fp32 addend written and read back) but ONE MORE K TILE of that layer's
    # recurrent ring-GEMM step (tacorl_rnn_linear_fwd_batch_ext); what remains in front is a 1 MB bf16 copy of the input rows.
    # TACORL_AD_PROJ_RING=0 (or proj_in_ring = False): the separate projection launch, as before.
    proj_in_ring = os.environ.get("TACORL_AD_PROJ_RING", "1") == "1"

    def _ring_proj(self):
        return bool(self.proj_in_ring) and os.environ.get("TACORL_AD_PROJ_RING", "1") == "1" and self.P + self.E <= 128

    def twin_input_proj(self, plan, emb, ld_emb, B, T, Tm):
        """Layer-0 input projection of the twin pass (reads the fp32 weights: may run before forward(), on another stream)."""
        tw, blk = self.twin_state(B, Tm), self.blk
        if self._ring_proj():  # only the bf16 input rows: the projection rides in the recurrent step's launch
            call("tacorl_build_ad_input_bf16", ptr(plan), ptr(emb), ld_emb, ptr(tw.xb_seq), B, T, Tm, self.P, self.E, ops.stream())
            return tw
        call("tacorl_ad_input_proj", ptr(plan), ptr(emb), ld_emb, blk.p("rnn.weight_ih_l0"), blk.p("rnn.bias_ih_l0"),
             ptr(tw.xin[0]), B, T, Tm, self.P, self.E, self.hidden, ops.stream())
        return tw

    def refresh_mirrors(self, B, Tm):
        """bf16 copies of the recurrent / projection / head weights for the ring GEMMs (weights only: forward() issues this
        itself unless told the mirrors are current; a caller may issue it early on another stream)."""
        self._ensure(B, Tm)
        blk, H = self.blk, self.hidden
        srcs = ([blk.p(f"rnn.weight_hh_l{l}") for l in range(self.L)] + [blk.p(f"rnn.weight_ih_l{l}") for l in range(1, self.L)]
                + [blk.p("mean_fc.weight")])
        dsts = self.whb + self.wib[1:] + [self.headw_b]
        call("tacorl_to_bf16_batch", len(srcs), ops.ptr_array(srcs), ops.ptr_array(dsts),
             (C.c_long * len(srcs))(*([H * H] * (len(srcs) - 1) + [self.NH * H])), ops.stream())
        self._mirror_ring = self._ring_proj()
        if self._mirror_ring:
            call("tacorl_pad_to_bf16", blk.p("rnn.weight_ih_l0"), self.P + self.E, ptr(self.wih0_b), 128, H, self.P + self.E, ops.stream())
        ob = blk.off["mean_fc.bias"][0]  # the four heads' biases sit back to back
        self.headb[: self.NH].copy_(blk.param[ob: ob + self.NH])

    def prepare_backward(self, B, Tm, compute):
        """Weights-only preparation of backward() (transposed bf16 copies for the BPTT ring GEMMs and the heads' input
        gradient); backward(prepared=True) then skips it.  May run early, on another stream."""
        if not self._bwd_fast(B, compute):
            return False
        self._ensure(B, Tm)
        self._ensure_bptt(B, Tm)
        self._transpose_weights()
        if self._heads_ring(B * Tm):
            call("tacorl_transpose_pad_to_bf16", self.blk.p("mean_fc.weight"), ptr(self.headwt_b), self.NH, self.hidden,
                 self.headwt_b.shape[1], ops.stream())
        return True

    def _bwd_fast(self, B, compute):
        H = self.hidden
        return compute == ops.BF16 and bool(ops.L.lib().tacorl_rnn_linear_supported(B, H, H)) and H % 32 == 0

    def _heads_ring(self, R):
        KP = (self.NH + 127) // 128 * 128  # the heads' width as a contraction length of the ring GEMM
        ok = getattr(self, "heads_dgrad_ring", True) and bool(ops.L.lib().tacorl_rnn_linear_supported(R, KP, self.hidden))
        if ok and getattr(self, "_hd_shape", None) != (R, KP):
            ops.note_alloc()
            self.d_heads_b = torch.zeros((R + 63) // 64 * 64, KP, device=self.dev, dtype=torch.bfloat16)  # (pad rows stay zero)
            self.headwt_b = torch.zeros(self.hidden, KP, device=self.dev, dtype=torch.bfloat16)
            self._hd_shape = (R, KP)
        return ok

    def _ensure_bptt(self, B, Tm):
        if getattr(self, "_bptt_shape", None) != (B, Tm):
            ops.note_alloc()
            H, L, R = self.hidden, self.L, B * Tm
            bf = lambda *s: torch.zeros(*s, device=self.dev, dtype=torch.bfloat16)  # noqa: E731
            self.DZb = [bf(R, H) for _ in range(L)]   # bf16 copies of dZ_t: the ring GEMM's operand
            self.whtb = [bf(H, H) for _ in range(L)]  # W_hh^T
            self.wihtb = [None] + [bf(H, H) for _ in range(1, L)]  # W_ih^T of layers >= 1 (wavefront projections)
            self._bptt_shape = (B, Tm)

    def twin_heads(self, twin, B, Tm):
        """Output heads of the twin pass (after forward(twin=...)): a launch of its own, for the caller's side stream."""
        if getattr(twin, "heads_done", False):
            return
        call("tacorl_rnn_linear_fwd", ptr(twin.hb[self.L - 1]), ptr(self.headw_b), ptr(self.headb), None, 0, ptr(twin.heads), None,
             B * Tm, self.hidden, self.NHP, ACT_NONE, ops.stream())
        twin.heads_done = True

    def _lin(self, x, ldx, w, b, y, M, K, N, act, compute, ldy=None):
        # split-K capable entry (skinny outputs with a long K: linear2 2048->32, the 2048->182 heads)
        nb = ops.L.lib().tacorl_linear_add_fwd_ws_bytes(1, ops.int_array([M]), K, N)
        ws = ops.workspace(nb, self.dev, "ad_splitk")  # own scratch: runs beside the plan recognition's linears
        call("tacorl_linear_add_fwd", 1, ops.ptr_array([x]), ldx, ops.ptr_array([w]), ops.ptr_array([b]), None, 0,
             ops.ptr_array([y]), N if ldy is None else ldy, ops.int_array([M]), K, N, act, compute, ptr(ws), ws.numel(),
             ops.stream())

    def forward(self, plan, emb, ld_emb, B, T, Tm, compute, frozen=False, mirrors_current=False, twin=None):
        """plan (B,P); emb [B*T][ld_emb] batch-major frame embeddings; uses steps t < Tm.  Fills self.heads.
        frozen=True: the caller never steps these weights with the library's optimiser kernels, so the bf16
        copies of the weight matrices are refreshed only when the block's torch version counter moved
        (load_state_dict / copy_).
        twin: a twin_state() whose layer-0 projection has been issued (twin_input_proj) - its recurrent steps and heads
        are computed as extra rows of this pass's launches (twin_ok() must hold); fills twin.heads."""
        self._ensure(B, Tm)
        blk, H, R = self.blk, self.hidden, B * Tm
        x, K = self.x_seq, self.P + self.E
        # bf16 mode: the layer-0 projection reads (plan, embeddings) itself (tacorl_ad_input_proj); x_seq is then only the
        # weight-gradient operand of a backward that follows (not built for the frozen, logging-only pass)
        # bf16 mode: the recurrent step runs as ONE launch (LDS-DMA ring GEMM, rnn_ops.hip) on bf16 copies of
        # W_hh (refreshed here: the weights may have been stepped) and of the previous hidden state
        fast = compute == ops.BF16 and bool(ops.L.lib().tacorl_rnn_linear_supported(B, H, H))
        # (the fused projection exists only on the ring-GEMM path: with a hidden size that path does not take - H % 128 != 0 -
        # the generic per-step path below reads x_seq, so it must be built; 32 <= K: the kernel's first k-step is unmasked)
        proj_fused = (fast and 32 <= K <= 64 and K % 8 == 0 and H % 16 == 0 and getattr(self, "fused_input_proj", True))
        # (the K extension of the ring GEMM takes input widths up to 128: two cameras + a 32-wide plan = 96 at C4, where the
        # separate projection was a generic GEMM launch at the head of the step's critical branch)
        ring = fast and self._ring_proj() and getattr(self, "fused_input_proj", True) and (proj_fused or twin is None)
        if not ((proj_fused or ring) and frozen):
            call("tacorl_build_ad_input", ptr(plan), ptr(emb), ld_emb, ptr(self.x_seq), B, T, Tm, self.P, self.E,
                 ops.stream())
        ver = blk.param._version
        # bf16 weight copies still valid: frozen weights at an unchanged version, or the caller has just run another forward
        # on these weights (PlayLMP: the logging-only random-plan pass and the real pass of one step - mirrors_current)
        fresh = (frozen and getattr(self, "_bf16_version", None) == ver) or (mirrors_current and getattr(self, "_shape", None) == (B, Tm))
        if fast:
            self._bf16_version = ver if frozen else None
        if fast and getattr(self, "_mirror_ring", None) != self._ring_proj():  # (the padded W_ih_l0 mirror exists only in ring mode)
            fresh = False
        if fast and not fresh:
            self.refresh_mirrors(B, Tm)
        at = ops._at
        if fast:
            # Wavefront over (layer, step): launch s holds the recurrent step s-2l of every layer l and the
            # input projection of step s-2l+1 of layers l >= 1 (operand h_{l-1}[s-2l+1] left launch s-1) as
            # independent problems of ONE batched ring-GEMM launch whose workgroups are co-resident:
            # T + 2(L-1) dependent launches instead of L*T + (L-1).
            if ring:
                call("tacorl_build_ad_input_bf16", ptr(plan), ptr(emb), ld_emb, ptr(self.xb_seq), B, T, Tm, self.P, self.E, ops.stream())
            elif proj_fused:
                call("tacorl_ad_input_proj", ptr(plan), ptr(emb), ld_emb, blk.p("rnn.weight_ih_l0"), blk.p("rnn.bias_ih_l0"),
                     ptr(self.xin[0]), B, T, Tm, self.P, self.E, H, ops.stream())
            else:
                self._lin(x, K, blk.p("rnn.weight_ih_l0"), blk.p("rnn.bias_ih_l0"), self.xin[0], R, K, H, ACT_NONE, compute)
            L = self.L
            assert twin is None or (proj_fused and twin.shape == (B, Tm)), "twin pass: twin_ok() / twin_state(B, Tm)"
            hbp = lambda l, t, o=self: C.c_void_p(o.hb[l].data_ptr() + 2 * t * B * H)  # noqa: E731
            xbp = lambda t, o=self: C.c_void_p(o.xb_seq.data_ptr() + 2 * t * B * 128)  # noqa: E731
            for s_ in range(Tm + 2 * (L - 1)):
                xs, wt, bs, ad, ys, yb, ac = [], [], [], [], [], [], []
                x2, ad2, y2, yb2 = [], [], [], []
                xe, xe2, we, b2 = [], [], [], []  # K extension (ring): layer 0's input rows / W_ih_l0 / b_ih_l0
                for l in range(L):
                    t = s_ - 2 * l
                    if 0 <= t < Tm:  # h_l[t] = relu(W_hh h_l[t-1] + b_hh + xin_l[t])
                        ext = ring and l == 0  # ... with xin_0[t] = W_ih x_t + b_ih as one more K tile of this problem
                        xs.append(ptr(self.h0b) if t == 0 else hbp(l, t - 1)); wt.append(ptr(self.whb[l]))
                        bs.append(blk.p(f"rnn.bias_hh_l{l}")); ad.append(None if ext else at(self.xin[l], t * B * H))
                        ys.append(at(self.h[l], t * B * H)); yb.append(hbp(l, t)); ac.append(ACT_RELU)
                        xe.append(xbp(t) if ext else None); we.append(ptr(self.wih0_b) if ext else None)
                        b2.append(blk.p("rnn.bias_ih_l0") if ext else None)
                        if twin is not None:
                            x2.append(ptr(self.h0b) if t == 0 else hbp(l, t - 1, twin))
                            ad2.append(None if ext else at(twin.xin[l], t * B * H))
                            y2.append(at(twin.h[l], t * B * H)); yb2.append(hbp(l, t, twin))
                            xe2.append(xbp(t, twin) if ext else None)
                    t = s_ - 2 * l + 1
                    if l >= 1 and 0 <= t < Tm:  # xin_l[t] = W_ih h_{l-1}[t] + b_ih
                        xs.append(hbp(l - 1, t)); wt.append(ptr(self.wib[l])); bs.append(blk.p(f"rnn.bias_ih_l{l}"))
                        ad.append(None); ys.append(at(self.xin[l], t * B * H)); yb.append(None); ac.append(ACT_NONE)
                        xe.append(None); we.append(None); b2.append(None)
                        if twin is not None:
                            x2.append(hbp(l - 1, t, twin)); ad2.append(None); y2.append(at(twin.xin[l], t * B * H)); yb2.append(None)
                            xe2.append(None)
                if ring:
                    tw_ = twin is not None
                    call("tacorl_rnn_linear_fwd_batch_ext", len(xs), ops.ptr_array(xs), ops.ptr_array(x2) if tw_ else None,
                         ops.ptr_array(wt), ops.ptr_array(bs), ops.ptr_array(ad), ops.ptr_array(ad2) if tw_ else None, H,
                         ops.ptr_array(ys), ops.ptr_array(y2) if tw_ else None, ops.ptr_array(yb), ops.ptr_array(yb2) if tw_ else None,
                         B, B if tw_ else 0, H, H, ops.int_array(ac), ops.ptr_array(xe), ops.ptr_array(xe2) if tw_ else None,
                         ops.ptr_array(we), ops.ptr_array(b2), ops.stream())
                    continue
                if twin is not None:
                    call("tacorl_rnn_linear_fwd_batch_twin", len(xs), ops.ptr_array(xs), ops.ptr_array(x2), ops.ptr_array(wt),
                         ops.ptr_array(bs), ops.ptr_array(ad), ops.ptr_array(ad2), H, ops.ptr_array(ys), ops.ptr_array(y2),
                         ops.ptr_array(yb), ops.ptr_array(yb2), B, B, H, H, ops.int_array(ac), ops.stream())
                    continue
                call("tacorl_rnn_linear_fwd_batch", len(xs), ops.ptr_array(xs), ops.ptr_array(wt), ops.ptr_array(bs),
                     ops.ptr_array(ad), H, ops.ptr_array(ys), ops.ptr_array(yb), B, H, H, ops.int_array(ac), ops.stream())
            x, K = self.h[L - 1], H
        for l in range(self.L if fast else 0, self.L):  # exact / generic path: layer by layer, step by step
            self._lin(x, K, blk.p(f"rnn.weight_ih_l{l}"), blk.p(f"rnn.bias_ih_l{l}"), self.xin[l], R, K, H, ACT_NONE,
                      compute)
            for t in range(Tm):
                prev = self.h0 if t == 0 else at(self.h[l], (t - 1) * B * H)
                call("tacorl_linear_add_fwd", 1, ops.ptr_array([prev]), H, ops.ptr_array([blk.p(f"rnn.weight_hh_l{l}")]),
                     ops.ptr_array([blk.p(f"rnn.bias_hh_l{l}")]), ops.ptr_array([at(self.xin[l], t * B * H)]), H,
                     ops.ptr_array([at(self.h[l], t * B * H)]), H, ops.int_array([B]), H, H, ACT_RELU, compute,
                     ptr(self.rnn_ws), self.rnn_ws.numel(), ops.stream())
            x, K = self.h[l], H
        if fast:  # output heads through the ring GEMM (bf16 weights, rows padded to a multiple of 32)
            if twin is not None and not getattr(self, "twin_heads_apart", True):
                call("tacorl_rnn_linear_fwd_batch_twin", 1, ops.ptr_array([self.hb[self.L - 1]]), ops.ptr_array([twin.hb[self.L - 1]]),
                     ops.ptr_array([self.headw_b]), ops.ptr_array([self.headb]), None, None, 0, ops.ptr_array([self.heads]),
                     ops.ptr_array([twin.heads]), None, None, R, R, H, self.NHP, ops.int_array([ACT_NONE]), ops.stream())
                twin.heads_done = True
                return
            if twin is not None:
                twin.heads_done = False  # (the caller issues twin_heads() where it wants them: off the real pass's chain)
            call("tacorl_rnn_linear_fwd", ptr(self.hb[self.L - 1]), ptr(self.headw_b), ptr(self.headb), None, 0,
                 ptr(self.heads), None, R, H, self.NHP, ACT_NONE, ops.stream())
        else:
            self._lin(x, H, blk.p("mean_fc.weight"), blk.p("mean_fc.bias"), self.heads, R, H, self.NH, ACT_NONE, compute,
                      ldy=self.NHP)

    def loss(self, actions, loss_out, B, T, Tm, want_grad, grad_scale=1.0, twin=None, lazy=False):
        """actions: device [B][T][Da+1]; writes the scalar loss to loss_out (device float) and, if
        want_grad, dL/dheads into self.d_heads.  twin: the loss of that twin pass's heads instead (no gradient).
        lazy: the scalar is only logged - leave the per-block partial sums in the workspace; finish_loss() (the module calls it
        before it reads the logs) sums them into loss_out.  One launch fewer at the end of the step's decoder branch."""
        heads, ws = (self.heads, self.ws) if twin is None else (twin.heads, twin.ws)
        call("tacorl_logistic_mixture_loss", ptr(heads), self.NHP, ptr(actions), ptr(self.d_heads) if want_grad and twin is None else None,
             None if lazy else loss_out, B, T, Tm, self.Da, self.K, self.num_classes, float(self.gripper_alpha), float(grad_scale),
             ptr(ws), ws.numel(), ops.stream())
        self._lazy_loss = (ws, B, Tm, loss_out) if lazy else None

    def finish_loss(self):
        """Sum the partials a lazy loss() left behind into its log slot (no-op otherwise)."""
        pend = getattr(self, "_lazy_loss", None)
        if pend is not None:
            ws, B, Tm, loss_out = pend
            call("tacorl_logistic_mixture_finish", ptr(ws), ws.numel(), B, Tm, self.Da, loss_out, ops.stream())

    def loss_step(self, module, actions, plan, B, T, optimize, frozen=False, defer_update=False, mirrors_current=False,
                  prepared=False):
        """TACORL.compute_action_decoder_update (reference tacorl.py:206-233): loss on emb[:, :-1],
        actions[:, :-1]; logged always, Adam step when fine-tuning."""
        from .._lib import LOG_SLOTS

        acts = actions
        cams = module.action_decoder_modalities
        if len(cams) == 1:  # one camera: the frame embeddings as the encoder wrote them
            emb, ld = module.f_out[cams[0]], 32
        elif cams == module.plan_recognition_modalities:  # already concatenated for the plan recognition
            emb, ld = module.pr_in, module.pr_in.shape[1]
        else:
            if getattr(module, "_ad_in", None) is None or module._ad_in.shape[0] != B * T:
                ops.note_alloc()
                module._ad_in = torch.zeros(B * T, 32 * len(cams), device=self.dev)
            for j, c in enumerate(cams):
                ops.copy_cols(module.f_out[c], 0, 32, module._ad_in, 32 * j, module._ad_in.shape[1], B * T, 32)
            emb, ld = module._ad_in, module._ad_in.shape[1]
        # (mirrors_current / prepared: the caller has issued refresh_mirrors() / prepare_backward() for the current weights)
        self.forward(plan, emb, ld, B, T, T - 1, module.compute, frozen=frozen and not optimize, mirrors_current=mirrors_current)
        slot = ops._at(module.engine.logs, LOG_SLOTS.index("action_loss"))
        self.loss(acts, slot, B, T, T - 1, want_grad=optimize, grad_scale=1.0 / module.world_size,
                  lazy=bool(getattr(module, "ad_loss_lazy", True)) and os.environ.get("TACORL_AD_LOSS_LAZY", "1") == "1")
        if optimize:
            # (no wgrad_stream here.  This call already runs on a branch of the step's graph: a side stream joined back INTO
            # that branch crashed hipStreamEndCapture on ROCm 7.2; forked from and joined into the main stream instead - with
            # the Adam step moved behind the join - it captured, and the step went 1.82 -> 2.10 ms: three concurrent
            # chains of chip-wide kernels slow each other more than the overlap returns)
            # (round 4: BPTT as the wavefront of batched launches here too.  Round 2 measured it 35 % slower beside the CQL
            # update; with the step as it is now - small-tile head / tail launches, one-launch RNN weight gradients - it is
            # faster: C3 1.683 -> 1.571 ms/step, its B = 32 share 1.078 -> 0.903, same-process A/B)
            self.backward(B, T - 1, module.compute, need_input_grad=False, wavefront=getattr(self, "finetune_wavefront", True),
                          prepared=prepared)
            # defer_update (more than one GPU): the gradient block lives in the engine's arena and is reduced by the
            # step's second collective; the module steps it afterwards (update())
            if not defer_update:
                self.update(module)

    def update(self, module):
        ops.adam_step(self.blk.param, self.blk.grad, self.blk.m, self.blk.v, module.action_decoder_lr, 0.0, self.blk.step)

    # ------------------------------------------------------------------ backward (BPTT)
    def _wgrad(self, x, ldx, dz, ld_dz, M, K, O, dw, db, compute):
        nb = ops.L.lib().tacorl_linear_wgrad_ws_bytes(1, ops.int_array([M]), K, O)
        ws = ops.workspace(nb, self.dev, "ad_wgrad")  # own scratch (see _lin)
        call("tacorl_linear_wgrad", 1, ops.ptr_array([x]), ldx, ops.ptr_array([dz]), ld_dz, ops.int_array([M]), K, O,
             ops.ptr_array([dw]), ops.ptr_array([db]) if db is not None else None, 0, compute, ptr(ws), ws.numel(),
             ops.stream())

    def _dgrad(self, dz, ld_dz, w, out, ld_out, M, O, I, compute, src=None, ld_src=0, act=ACT_NONE, addend=None,
               ld_add=0):
        nb = ops.L.lib().tacorl_linear_dgrad_ws_bytes(1, ops.int_array([M]), O, I)  # split reduction for skinny outputs
        ws = ops.workspace(nb, self.dev, "ad_dgrad")
        call("tacorl_linear_dgrad_splitk", 1, ops.ptr_array([dz]), ld_dz, ops.ptr_array([w]), ops.ptr_array([out]), ld_out,
             ops.ptr_array([src]) if src is not None else None, ld_src, act,
             ops.ptr_array([addend]) if addend is not None else None, ld_add, ops.int_array([M]), O, I, compute,
             ptr(ws), ws.numel(), ops.stream())

    def _square_wgrads(self, B, Tm):
        """The square matrices' gradients of ALL layers (W_hh_l, W_ih_l for l >= 1, the latter with the bias) in one launch
        (rnn_ops.hip tacorl_rnn_wgrad_batch; 2 L - 1 launches before), the bias copies behind it.  Returns False when the shapes
        do not qualify - the caller then goes layer by layer."""
        blk, H, R, L = self.blk, self.hidden, B * Tm, self.L
        sup = ops.L.lib().tacorl_rnn_wgrad_supported
        if (os.environ.get("TACORL_AD_WGRAD_BATCH", "1") != "1" or not getattr(self, "wgrad_batched", True) or Tm < 2
                or 2 * L - 1 > 4 or L < 2 or not sup((Tm - 1) * B, H, H) or not sup(R, H, H)):
            return False
        bfp = lambda t, off: C.c_void_p(t.data_ptr() + 2 * off)  # noqa: E731
        dz, x, rows, dw, db = [], [], [], [], []
        for l in range(L):
            dz.append(bfp(self.DZb[l], B * H)); x.append(ptr(self.hb[l])); rows.append((Tm - 1) * B)
            dw.append(blk.g(f"rnn.weight_hh_l{l}")); db.append(None)
        for l in range(1, L):
            dz.append(ptr(self.DZb[l])); x.append(ptr(self.hb[l - 1])); rows.append(R)
            dw.append(blk.g(f"rnn.weight_ih_l{l}")); db.append(blk.g(f"rnn.bias_ih_l{l}"))
        call("tacorl_rnn_wgrad_batch", len(dz), ops.ptr_array(dz), H, ops.ptr_array(x), H, ops.int_array(rows), H, H,
             ops.ptr_array(dw), ops.ptr_array(db), 0, ops.stream())
        return True

    def _layer_wgrads(self, l, B, Tm, compute, fast, square_done=False):
        """W_hh / W_ih / bias gradients of layer l (its BPTT has been issued)."""
        blk, H, R = self.blk, self.hidden, B * Tm
        if square_done:  # _square_wgrads has the square matrices: layer 0's input matrix and the bias copies are left
            if l == 0:
                self._wgrad(self.x_seq, self.P + self.E, self.DZ[0], H, R, self.P + self.E, H, blk.g("rnn.weight_ih_l0"),
                            blk.g("rnn.bias_ih_l0"), compute)
            call("tacorl_copy_cols", blk.g(f"rnn.bias_ih_l{l}"), H, blk.g(f"rnn.bias_hh_l{l}"), H, 1, H, 0, 0, ops.stream())
            return
        h, DZ, at = self.h[l], self.DZ[l], ops._at
        # the square matrices' gradients straight from the bf16 copies the ring GEMMs left (hb: forward, DZb: BPTT):
        # one launch per matrix, no slabs (rnn_ops.hip rnn_wgrad_kernel); other shapes: the generic split-R GEMM
        tr = lambda rows: fast and bool(ops.L.lib().tacorl_rnn_wgrad_supported(rows, H, H))  # noqa: E731
        bfp = lambda t, off: C.c_void_p(t.data_ptr() + 2 * off)  # noqa: E731
        if Tm > 1 and tr((Tm - 1) * B):
            call("tacorl_rnn_wgrad", bfp(self.DZb[l], B * H), H, ptr(self.hb[l]), H, (Tm - 1) * B, H, H,
                 blk.g(f"rnn.weight_hh_l{l}"), None, 0, ops.stream())
        elif Tm > 1:
            self._wgrad(h, H, at(DZ, B * H), H, (Tm - 1) * B, H, H, blk.g(f"rnn.weight_hh_l{l}"), None, compute)
        else:
            blk.grad_views[f"rnn.weight_hh_l{l}"].zero_()
        xin, K = (self.x_seq, self.P + self.E) if l == 0 else (self.h[l - 1], H)
        if l > 0 and Tm > 1 and tr(R):
            call("tacorl_rnn_wgrad", ptr(self.DZb[l]), H, ptr(self.hb[l - 1]), H, R, H, H, blk.g(f"rnn.weight_ih_l{l}"),
                 blk.g(f"rnn.bias_ih_l{l}"), 0, ops.stream())
        else:
            self._wgrad(xin, K, DZ, H, R, K, H, blk.g(f"rnn.weight_ih_l{l}"), blk.g(f"rnn.bias_ih_l{l}"), compute)
        call("tacorl_copy_cols", blk.g(f"rnn.bias_ih_l{l}"), H, blk.g(f"rnn.bias_hh_l{l}"), H, 1, H, 0, 0,
             ops.stream())

    def _bptt_wavefront(self, B, Tm, prepared=False):
        """BPTT of all layers as a wavefront of batched ring-GEMM launches (rnn_ops.hip tacorl_rnn_linear_bwd_batch).
        With k = L-1-l the depth of layer l below the top, launch s holds
          step(l, u), u = Tm-2-(s-2k):  dZ_l[u] = (dZ_l[u+1] W_hh_l + dH_l[u]) * [h_l[u] > 0]
          proj(l, t), t = Tm-1-(s-2k):  dH_{l-1}[t] = dZ_l[t] W_ih_l  (t = Tm-1: masked by h_{l-1}[t], i.e. dZ_{l-1}[Tm-1] itself)
        whose operands all left launch s-1: Tm-1 + 2(L-1) launches instead of L(Tm-1) + (L-1) projection GEMMs."""
        blk, H, L = self.blk, self.hidden, self.L
        at = ops._at
        bfp = lambda t, off: C.c_void_p(t.data_ptr() + 2 * off)  # noqa: E731
        row = lambda t: t * B * H  # noqa: E731
        if not prepared:
            self._transpose_weights()
        top, last = L - 1, row(Tm - 1)
        call("tacorl_relu_mask_mul", at(self.dHs[top], last), None, at(self.h[top], last), at(self.DZ[top], last), B * H, ops.stream())
        call("tacorl_to_bf16_batch", 1, ops.ptr_array([at(self.DZ[top], last)]), ops.ptr_array([bfp(self.DZb[top], last)]),
             (C.c_long * 1)(B * H), ops.stream())
        for s_ in range(Tm - 1 + 2 * (L - 1)):
            xs, wt, ad, ms, ys, yb = [], [], [], [], [], []
            for l in range(L):
                j = s_ - 2 * (L - 1 - l)
                if j < 0:
                    continue
                u = Tm - 2 - j
                if u >= 0:
                    xs.append(bfp(self.DZb[l], row(u + 1))); wt.append(ptr(self.whtb[l])); ad.append(at(self.dHs[l], row(u)))
                    ms.append(at(self.h[l], row(u))); ys.append(at(self.DZ[l], row(u))); yb.append(bfp(self.DZb[l], row(u)))
                t = Tm - 1 - j
                if l >= 1 and t >= 0:
                    xs.append(bfp(self.DZb[l], row(t))); wt.append(ptr(self.wihtb[l])); ad.append(None)
                    if t == Tm - 1:
                        ms.append(at(self.h[l - 1], row(t))); ys.append(at(self.DZ[l - 1], row(t))); yb.append(bfp(self.DZb[l - 1], row(t)))
                    else:
                        ms.append(None); ys.append(at(self.dHs[l - 1], row(t))); yb.append(None)
            if xs:
                call("tacorl_rnn_linear_bwd_batch", len(xs), ops.ptr_array(xs), ops.ptr_array(wt), ops.ptr_array(ad), H,
                     ops.ptr_array(ms), ops.ptr_array(ys), ops.ptr_array(yb), B, H, H, ops.stream())

    def _transpose_weights(self):
        blk, H = self.blk, self.hidden
        srcs = [blk.p(f"rnn.weight_hh_l{l}") for l in range(self.L)] + [blk.p(f"rnn.weight_ih_l{l}") for l in range(1, self.L)]
        dsts = [self.whtb[l] for l in range(self.L)] + [self.wihtb[l] for l in range(1, self.L)]
        for i in range(0, len(srcs), 16):  # one launch (round 5: 2 L - 1 before)
            n = len(srcs[i: i + 16])
            call("tacorl_transpose_to_bf16_batch", n, ops.ptr_array(srcs[i: i + 16]), ops.ptr_array(dsts[i: i + 16]),
                 ops.int_array([H] * n), ops.int_array([H] * n), ops.stream())

    def backward(self, B, Tm, compute, need_input_grad=False, wgrad_stream=None, join=True, wavefront=True, prepared=False):
        """Gradients of the loss (dL/dheads in self.d_heads) into self.blk.grad; optionally
        d(x_seq) into self.dx_seq.  ReLU-RNN BPTT: dz_{t-1} = (dz_t W_hh + dH_{t-1}) * [h_{t-1} > 0].
        wgrad_stream: the weight gradients - which only the optimiser reads - go to that stream (a branch of a
        captured graph) beside the dependent chain heads -> BPTT -> input gradient; join=False leaves the join
        (`current.wait_stream(wgrad_stream)`) to the caller, who may put more work in front of it.
        wavefront: BPTT as Tm+1 batched launches (_bptt_wavefront) instead of one launch per (layer, step) and a projection
        GEMM per layer (PlayLMP: 3.06 -> 2.98 ms/step in round 2; TACORL fine-tuning, a branch beside the CQL update: slower in
        round 2 - 1.80 -> 2.43 - and faster in round 4 - 1.68 -> 1.57 -, see loss_step)."""
        blk, H, R, L = self.blk, self.hidden, B * Tm, self.L
        if getattr(self, "_bshape", None) != (B, Tm):
            ops.note_alloc()
            f = lambda *s: torch.zeros(*s, device=self.dev)  # noqa: E731
            self.dHs = [f(R, H) for _ in range(L)]  # dL/dh_l before the ReLU mask, per layer (the wavefront needs them side by side)
            self.dH = self.dHs[L - 1]
            self.DZ = [f(R, H) for _ in range(L)]
            self.dx_seq = f(R, self.P + self.E)
            self._bshape = (B, Tm)
        at = ops._at

        def side(fn):  # everything fn reads has been issued on the current stream; nothing on that stream reads fn's outputs
            if wgrad_stream is None:
                return fn()
            wgrad_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(wgrad_stream):
                fn()

        fast = self._bwd_fast(B, compute)
        ring = fast and self._heads_ring(R)
        if ring:
            # dH = d_heads W through the ring GEMM (K = 182 padded to 256, two k-steps): the generic GEMM spends 64 us at
            # 3 840 rows on this K-short, epilogue-bound product.  Operands: d_heads as K-padded bf16, W^T K-padded.
            KP = self.headwt_b.shape[1]
            if not prepared:
                call("tacorl_transpose_pad_to_bf16", blk.p("mean_fc.weight"), ptr(self.headwt_b), self.NH, H, KP, ops.stream())
            call("tacorl_pad_to_bf16", ptr(self.d_heads), self.NHP, ptr(self.d_heads_b), KP, R, self.NH, ops.stream())
        # the heads' weight gradient [182][H] = d_heads^T h: from the same bf16 operands through the transposing-read kernel
        # of the recurrent matrices, the rows cut into slabs that run side by side (2 x 16 output tiles alone would leave the
        # chip idle); generic split GEMM + slab reduce otherwise (97 us at 3 840 rows)
        Rp = (R + 63) // 64 * 64
        slabs = max(d for d in range(1, 9) if (Rp // 64) % d == 0)
        if (ring and getattr(self, "heads_wgrad_slabs", True) and H % 128 == 0
                and bool(ops.L.lib().tacorl_rnn_wgrad_supported(Rp // slabs, self.headwt_b.shape[1], H))):
            def heads_wgrad():
                KP = self.headwt_b.shape[1]
                nb = ops.L.lib().tacorl_rnn_wgrad_slabs_ws_bytes(slabs, KP, H)
                ws = ops.workspace(nb, self.dev, "ad_heads_wgrad")
                call("tacorl_rnn_wgrad_slabs", ptr(self.d_heads_b), KP, ptr(self.hb[L - 1]), H, Rp, KP, H, self.NH, slabs,
                     blk.g("mean_fc.weight"), blk.g("mean_fc.bias"), 0, ptr(ws), ws.numel(), ops.stream())
            side(heads_wgrad)
        else:
            side(lambda: self._wgrad(self.h[L - 1], H, self.d_heads, self.NHP, R, H, self.NH, blk.g("mean_fc.weight"),
                                     blk.g("mean_fc.bias"), compute))
        if ring:
            call("tacorl_rnn_linear_bwd_batch", 1, ops.ptr_array([self.d_heads_b]), ops.ptr_array([self.headwt_b]), None, H, None,
                 ops.ptr_array([self.dH]), None, R, KP, H, ops.stream())
        else:
            self._dgrad(self.d_heads, self.NHP, blk.p("mean_fc.weight"), self.dH, H, R, self.NH, H, compute)
        if fast:
            self._ensure_bptt(B, Tm)
        if fast and Tm > 1 and 2 * L - 1 <= 4 and wavefront and getattr(self, "bptt_wavefront", True):
            self._bptt_wavefront(B, Tm, prepared)
            def all_wgrads():
                sq = self._square_wgrads(B, Tm)
                for l in reversed(range(L)):
                    self._layer_wgrads(l, B, Tm, compute, fast, sq)
            side(all_wgrads)
            if need_input_grad:
                K = self.P + self.E
                self._dgrad(self.DZ[0], H, blk.p("rnn.weight_ih_l0"), self.dx_seq, K, R, H, K, compute)
            if wgrad_stream is not None and join:
                torch.cuda.current_stream().wait_stream(wgrad_stream)
            return
        for l in reversed(range(L)):
            h, DZ = self.h[l], self.DZ[l]
            last = (Tm - 1) * B * H
            call("tacorl_relu_mask_mul", at(self.dHs[l], last), None, at(h, last), at(DZ, last), B * H, ops.stream())
            if fast and Tm > 1:
                # BPTT steps as one launch each (LDS-DMA ring GEMM on W_hh^T) instead of a 64-workgroup GEMM
                call("tacorl_transpose_to_bf16", blk.p(f"rnn.weight_hh_l{l}"), ptr(self.whtb[l]), H, H, ops.stream())
                call("tacorl_to_bf16_batch", 1, ops.ptr_array([at(DZ, last)]),
                     ops.ptr_array([C.c_void_p(self.DZb[l].data_ptr() + 2 * last)]), (C.c_long * 1)(B * H), ops.stream())
                for t in range(Tm - 1, 0, -1):
                    call("tacorl_rnn_linear_bwd_step", C.c_void_p(self.DZb[l].data_ptr() + 2 * t * B * H), ptr(self.whtb[l]),
                         at(self.dHs[l], (t - 1) * B * H), H, at(h, (t - 1) * B * H), at(DZ, (t - 1) * B * H),
                         C.c_void_p(self.DZb[l].data_ptr() + 2 * (t - 1) * B * H), B, H, H, ops.stream())
            else:
                for t in range(Tm - 1, 0, -1):
                    self._dgrad(at(DZ, t * B * H), H, blk.p(f"rnn.weight_hh_l{l}"), at(DZ, (t - 1) * B * H), H, B, H, H,
                                compute, src=at(h, (t - 1) * B * H), ld_src=H, act=ACT_RELU,
                                addend=at(self.dHs[l], (t - 1) * B * H), ld_add=H)
            side(lambda l=l: self._layer_wgrads(l, B, Tm, compute, fast))
            if l > 0:
                self._dgrad(DZ, H, blk.p(f"rnn.weight_ih_l{l}"), self.dHs[l - 1], H, R, H, H, compute)
            elif need_input_grad:
                K = self.P + self.E
                self._dgrad(DZ, H, blk.p("rnn.weight_ih_l0"), self.dx_seq, K, R, H, K, compute)
        if wgrad_stream is not None and join:
            torch.cuda.current_stream().wait_stream(wgrad_stream)
