"""PlanRecognitionTransformersNetwork on HIP kernels (reference
networks/plan_encoders/plan_recognition_transformer.py:10-105).

Parameters keep the reference's state-dict names (position_embeddings, layernorm,
transformer_encoder.layers.N.{self_attn.in_proj_*, self_attn.out_proj.*, linear1/2.*, norm1/2.*},
fc, mean_fc, variance_fc).  mean_fc and variance_fc are stored back to back so the two heads
are one GEMM.  `fc` is applied after the mean over time (fc is affine, so
mean_t(fc(x_t)) == fc(mean_t(x_t)); saves 16x of the 32->4096 GEMM).

Train mode (`forward(train=True)` with dropout_p > 0, reference :49-54,60,87 and
config/networks/plan_recognition/transformer.yaml:9): the 1 + 4 * num_layers dropout sites of the reference -
embeddings, and per encoder layer attention probabilities, dropout1, FFN dropout, dropout2 - take their keep
masks as explicit inputs (`stage_dropout`: injected masks in the reference's layouts, or fresh Bernoulli draws),
applied in the forward and to the matching gradients in the backward.
"""
import torch

from .. import ops
from .._lib import ACT_NONE, ACT_RELU, call, ptr
from ..blocks import TensorBlock


class PlanRecognition:
    def __init__(self, state_dim, latent_plan_dim, device, num_heads=8, num_layers=2, encoder_hidden_size=2048,
                 fc_hidden_size=4096, max_position_embeddings=16, min_std=1e-4, dropout_p=0.0, trainable=True,
                 **unused):
        self.D_in = state_dim
        self.pad = (-state_dim) % num_heads
        self.D = state_dim + self.pad
        self.A, self.H, self.L = latent_plan_dim, num_heads, num_layers
        self.FF, self.FC, self.T_max, self.min_std = encoder_hidden_size, fc_hidden_size, max_position_embeddings, min_std
        self.latent_plan_dim = latent_plan_dim
        self.dropout_p = dropout_p
        D, FF, FC, A = self.D, self.FF, self.FC, self.A
        spec = [("position_embeddings.weight", (self.T_max, D)), ("layernorm.weight", (D,)), ("layernorm.bias", (D,))]
        for l in range(num_layers):
            p = f"transformer_encoder.layers.{l}."
            spec += [(p + "self_attn.in_proj_weight", (3 * D, D)), (p + "self_attn.in_proj_bias", (3 * D,)),
                     (p + "self_attn.out_proj.weight", (D, D)), (p + "self_attn.out_proj.bias", (D,)),
                     (p + "linear1.weight", (FF, D)), (p + "linear1.bias", (FF,)),
                     (p + "linear2.weight", (D, FF)), (p + "linear2.bias", (D,)),
                     (p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
                     (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,))]
        spec += [("fc.weight", (FC, D)), ("fc.bias", (FC,)), ("mean_fc.weight", (A, FC)),
                 ("variance_fc.weight", (A, FC)), ("mean_fc.bias", (A,)), ("variance_fc.bias", (A,))]
        self.blk = TensorBlock(spec, device, trainable=trainable)
        self.dev = device
        self._shape = None

    def _ensure(self, B, T):
        if self._shape == (B, T):
            return
        if T > self.T_max:
            raise ValueError(f"sequence length {T} > max_position_embeddings {self.T_max}")
        ops.note_alloc()
        f = lambda *s: torch.zeros(*s, device=self.dev)  # noqa: E731
        R, D = B * T, self.D
        self.x = [f(R, D) for _ in range(2 * self.L + 1)]  # layer inputs / post-norm1 / post-norm2
        self.qkv = [f(R, 3 * D) for _ in range(self.L)]
        self.att = [f(R, D) for _ in range(self.L)]
        self.proj = [f(R, D) for _ in range(self.L)]
        self.ff1 = [f(R, self.FF) for _ in range(self.L)]
        self.ff2 = [f(R, D) for _ in range(self.L)]
        self.stats = [f(R, 2) for _ in range(2 * self.L)]
        self.pooled, self.fc_out, self.head = f(B, D), f(B, self.FC), f(B, 2 * self.A)
        self.keep = None
        if self.dropout_p > 0:  # keep masks (uint8): [embedding] + per layer [attention probs, dropout1, ffn, dropout2]
            u8 = lambda n: torch.ones(n, dtype=torch.uint8, device=self.dev)  # noqa: E731
            self.keep = [u8(R * D)]
            for _ in range(self.L):
                self.keep += [u8(B * self.H * T * T), u8(R * D), u8(R * self.FF), u8(R * D)]
        self._shape = (B, T)

    def stage_dropout(self, B, T, masks=None):
        """Fill the keep masks of the next train-mode forward (eager, before a graph replay - like the step's other
        noise).  masks: the reference's draws in its order and layouts (sequence-major (T,B,*) for the activations,
        (B,H,T,T) for the attention probabilities); None: fresh Bernoulli(1 - p) draws from torch's device generator."""
        self._ensure(B, T)
        if self.keep is None:
            return
        if masks is None:
            for k in self.keep:
                k.bernoulli_(1.0 - self.dropout_p)
            return
        assert len(masks) == len(self.keep), (len(masks), len(self.keep))
        for i, (k, m) in enumerate(zip(self.keep, masks)):
            m = m.to(self.dev)
            if (i - 1) % 4 != 0 or i == 0:  # activation masks arrive as (T, B, *): batch-major here
                m = m.permute(1, 0, 2)
            k.copy_(m.reshape(-1).to(torch.uint8))

    def _drop(self, x, i, n):
        call("tacorl_dropout_mul", ptr(x), ptr(self.keep[i]), 1.0 / (1.0 - self.dropout_p), n, ops.stream())

    def _lin(self, x, ldx, w, b, y, M, K, N, act, compute):
        # split-K capable entry (skinny outputs with a long K: linear2 2048->32, the 2048->182 heads)
        nb = ops.L.lib().tacorl_linear_add_fwd_ws_bytes(1, ops.int_array([M]), K, N)
        ws = ops.workspace(nb, self.dev, "lin_splitk")
        call("tacorl_linear_add_fwd", 1, ops.ptr_array([x]), ldx, ops.ptr_array([w]), ops.ptr_array([b]), None, 0,
             ops.ptr_array([y]), N, ops.int_array([M]), K, N, act, compute, ptr(ws), ws.numel(), ops.stream())

    def _fused_offsets(self):
        names = ["position_embeddings.weight"]
        for l in range(self.L):
            p = f"transformer_encoder.layers.{l}."
            names += [p + k for k in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                                      "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                                      "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")]
        return [self.blk.off[n][0] for n in names]

    def fused_inference_ok(self, T, ld_emb, compute):
        return bool(compute == ops.BF16 and self.pad == 0 and ld_emb % 4 == 0 and 2 * self.A <= 64
                    and ops.L.lib().tacorl_pr_encoder_fused_supported(self.D, T, self.H, self.FF, self.L))

    def prepare_inference(self):
        """Weight-only preparation of the single-launch inference path: the bf16 mirror of the parameter block
        and the composed posterior head (fc -> mean_fc is one affine map).  Depends on nothing a step computes,
        so a caller can issue it beside the image encoders instead of on the plan's dependent chain."""
        import ctypes as C
        blk = self.blk
        if getattr(self, "_pb", None) is None:
            ops.note_alloc()
            self._pb = torch.zeros(blk.param.numel(), device=self.dev, dtype=torch.bfloat16)
            self._foff = (C.c_long * (1 + 12 * self.L))(*self._fused_offsets())
            self._Wc = torch.zeros(2 * self.A, self.D, device=self.dev)
            self._bc = torch.zeros(2 * self.A, device=self.dev)
        n4 = blk.param.numel() // 4 * 4
        call("tacorl_to_bf16_batch", 1, ops.ptr_array([blk.param]), ops.ptr_array([self._pb]), (C.c_long * 1)(n4),
             ops.stream())
        call("tacorl_pr_head_compose", blk.p("fc.weight"), blk.p("fc.bias"), blk.p("mean_fc.weight"), blk.p("mean_fc.bias"),
             ptr(self._Wc), ptr(self._bc), self.D, self.FC, 2 * self.A, ops.stream())

    def forward(self, emb, ld_emb, B, T, compute, inference=False, sample=None, prepared=False, frozen=False, train=False):
        """emb: device tensor/pointer of [B*T][ld_emb] per-frame embeddings (first D_in columns used).
        Returns the (B, 2A) head buffer [mean | var_raw].  inference=True (frozen network, no backward
        follows): the encoder layers + time pooling run as one launch when the shape qualifies; with
        sample=(eps, plan) the posterior head and plan = tanh(mean + eps * std) ride in that launch too
        (prepared=True: prepare_inference() was already issued for the current weights; frozen=True: the caller
        never steps these weights with the library's optimiser kernels, so they only change through torch in-place
        ops - load_state_dict, copy_ - which bump the parameter block's version counter: the preparation is then
        re-issued only when that counter moved)."""
        self._ensure(B, T)
        blk, D, R = self.blk, self.D, B * T
        drop = self._dropping = bool(train and not inference and self.dropout_p > 0)
        ks = 1.0 / (1.0 - self.dropout_p) if drop else 1.0
        if inference and self.fused_inference_ok(T, ld_emb, compute):
            ver = blk.param._version
            if not prepared and not (frozen and getattr(self, "_prep_version", None) == ver):
                self.prepare_inference()
                self._prep_version = ver if frozen else None
            if sample is not None:
                eps, plan = sample
                call("tacorl_pr_encoder_fused_sample", ptr(emb), ld_emb, ptr(blk.param), ptr(self._pb), self._foff,
                     ptr(self.pooled), B, D, T, self.H, self.FF, self.L, ptr(self._Wc), ptr(self._bc), ptr(eps),
                     ptr(self.head), ptr(plan), self.A, float(self.min_std), ops.stream())
                return self.head
            call("tacorl_pr_encoder_fused", ptr(emb), ld_emb, ptr(blk.param), ptr(self._pb), self._foff, ptr(self.pooled),
                 B, D, T, self.H, self.FF, self.L, ops.stream())
            self._lin(self.pooled, D, blk.p("fc.weight"), blk.p("fc.bias"), self.fc_out, B, D, self.FC, ACT_NONE, compute)
            self._lin(self.fc_out, self.FC, blk.p("mean_fc.weight"), blk.p("mean_fc.bias"), self.head, B, self.FC,
                      2 * self.A, ACT_NONE, compute)
            return self.head
        if (not inference and not drop and getattr(self, "fused_train", True) and self.fused_inference_ok(T, ld_emb, compute)
                and ops.L.lib().tacorl_pr_encoder_fused_train_supported(self.D, T, self.H, self.FF, self.L)):
            # train mode without dropout (bf16): the encoder layers + time pooling as ONE launch that also writes what the
            # per-op backward below reads (layer inputs, q|k|v, attention / projection / FFN outputs, LayerNorm statistics)
            # in the per-op forward's layouts - ~17 dependent launches of PlayLMP.training_step's chain become one
            import ctypes as C
            if not prepared:
                if getattr(self, "_pb", None) is None:
                    ops.note_alloc()
                    self._pb = torch.zeros(blk.param.numel(), device=self.dev, dtype=torch.bfloat16)
                    self._foff = (C.c_long * (1 + 12 * self.L))(*self._fused_offsets())
                    self._Wc = torch.zeros(2 * self.A, self.D, device=self.dev)
                    self._bc = torch.zeros(2 * self.A, device=self.dev)
                call("tacorl_to_bf16_batch", 1, ops.ptr_array([blk.param]), ops.ptr_array([self._pb]),
                     (C.c_long * 1)(blk.param.numel() // 4 * 4), ops.stream())
            self._prep_version = None  # (whatever is current now will not be after the optimiser step that follows)
            save = []
            for l in range(self.L):
                save += [self.x[2 * l], self.qkv[l], self.att[l], self.proj[l], self.x[2 * l + 1], self.ff1[l], self.ff2[l],
                         self.stats[2 * l], self.stats[2 * l + 1]]
            self._fused_saved = (B, T)  # (backward may take the fused chain: every saved tensor is in the fused layout)
            # prepared (prepare_inference() issued for the current weights, e.g. early on a side stream): the posterior head
            # as ONE affine map and the plan sample ride in the launch, as in the inference path - fc (32 -> 4096), mean_fc
            # (4096 -> 2A, split-K + reduce) and the sample kernel leave the dependent chain; backward() takes d_pool =
            # d_head Wc inside its launch and computes fc_out / d_fc for the two weight gradients beside the chain
            self._composed = bool(prepared and sample is not None and 2 * self.A <= 64 and getattr(self, "composed_head", True))
            if self._composed:
                eps, plan = sample
                call("tacorl_pr_encoder_fused_train_sample", ptr(emb), ld_emb, ptr(blk.param), ptr(self._pb), self._foff,
                     ptr(self.pooled), B, D, T, self.H, self.FF, self.L, ops.ptr_array(save), ptr(self._Wc), ptr(self._bc),
                     ptr(eps), ptr(self.head), ptr(plan), self.A, float(self.min_std), ops.stream())
                return self.head
            call("tacorl_pr_encoder_fused_train", ptr(emb), ld_emb, ptr(blk.param), ptr(self._pb), self._foff, ptr(self.pooled),
                 B, D, T, self.H, self.FF, self.L, ops.ptr_array(save), ops.stream())
            self._lin(self.pooled, D, blk.p("fc.weight"), blk.p("fc.bias"), self.fc_out, B, D, self.FC, ACT_NONE, compute)
            self._lin(self.fc_out, self.FC, blk.p("mean_fc.weight"), blk.p("mean_fc.bias"), self.head, B, self.FC,
                      2 * self.A, ACT_NONE, compute)
            if sample is not None:
                eps, plan = sample
                call("tacorl_pr_sample", ptr(self.head), ptr(eps), ptr(plan), None, None, B, self.A, float(self.min_std),
                     ops.stream())
            return self.head
        self._fused_saved, self._composed = None, False
        call("tacorl_add_rows_bcast", ptr(emb), ld_emb, blk.p("position_embeddings.weight"), ptr(self.x[0]), R, T,
             self.D_in, D, ops.stream())
        if drop:
            self._drop(self.x[0], 0, R * D)
        for l in range(self.L):
            p = f"transformer_encoder.layers.{l}."
            xin, x1, x2 = self.x[2 * l], self.x[2 * l + 1], self.x[2 * l + 2]
            self._lin(xin, D, blk.p(p + "self_attn.in_proj_weight"), blk.p(p + "self_attn.in_proj_bias"), self.qkv[l],
                      R, D, 3 * D, ACT_NONE, compute)
            if drop:
                call("tacorl_attention_dropout_fwd", ptr(self.qkv[l]), ptr(self.att[l]), ptr(self.keep[1 + 4 * l]), ks, B, T,
                     D, self.H, ops.stream())
            else:
                call("tacorl_attention_fwd", ptr(self.qkv[l]), ptr(self.att[l]), B, T, D, self.H, ops.stream())
            self._lin(self.att[l], D, blk.p(p + "self_attn.out_proj.weight"), blk.p(p + "self_attn.out_proj.bias"),
                      self.proj[l], R, D, D, ACT_NONE, compute)
            if drop:
                self._drop(self.proj[l], 2 + 4 * l, R * D)
            call("tacorl_add_layernorm_fwd", ptr(xin), ptr(self.proj[l]), blk.p(p + "norm1.weight"),
                 blk.p(p + "norm1.bias"), ptr(x1), ptr(self.stats[2 * l]), R, D, 1e-5, ops.stream())
            self._lin(x1, D, blk.p(p + "linear1.weight"), blk.p(p + "linear1.bias"), self.ff1[l], R, D, self.FF,
                      ACT_RELU, compute)
            if drop:
                self._drop(self.ff1[l], 3 + 4 * l, R * self.FF)
            self._lin(self.ff1[l], self.FF, blk.p(p + "linear2.weight"), blk.p(p + "linear2.bias"), self.ff2[l], R,
                      self.FF, D, ACT_NONE, compute)
            if drop:
                self._drop(self.ff2[l], 4 + 4 * l, R * D)
            call("tacorl_add_layernorm_fwd", ptr(x1), ptr(self.ff2[l]), blk.p(p + "norm2.weight"),
                 blk.p(p + "norm2.bias"), ptr(x2), ptr(self.stats[2 * l + 1]), R, D, 1e-5, ops.stream())
        call("tacorl_mean_over_t", ptr(self.x[2 * self.L]), ptr(self.pooled), B, T, D, ops.stream())
        self._lin(self.pooled, D, blk.p("fc.weight"), blk.p("fc.bias"), self.fc_out, B, D, self.FC, ACT_NONE, compute)
        self._lin(self.fc_out, self.FC, blk.p("mean_fc.weight"), blk.p("mean_fc.bias"), self.head, B, self.FC,
                  2 * self.A, ACT_NONE, compute)
        if sample is not None:
            eps, plan = sample
            call("tacorl_pr_sample", ptr(self.head), ptr(eps), ptr(plan), None, None, B, self.A, float(self.min_std),
                 ops.stream())
        return self.head

    # ------------------------------------------------------------------------ backward
    def _wgrad(self, x, ldx, dz, ld_dz, M, K, O, dw, db, compute):
        nb = ops.L.lib().tacorl_linear_wgrad_ws_bytes(1, ops.int_array([M]), K, O)
        ws = ops.workspace(nb, self.dev, "lin_wgrad")
        call("tacorl_linear_wgrad", 1, ops.ptr_array([x]), ldx, ops.ptr_array([dz]), ld_dz, ops.int_array([M]), K, O,
             ops.ptr_array([dw]), ops.ptr_array([db]), 0, compute, ptr(ws), ws.numel(), ops.stream())

    def _wgrad_n(self, xs, ldx, dzs, ld_dz, M, K, O, dws, dbs, compute):
        """Several weight gradients of one shape (the same Linear of every transformer layer) as one launch + one reduce."""
        n = len(xs)
        nb = ops.L.lib().tacorl_linear_wgrad_ws_bytes(n, ops.int_array([M] * n), K, O)
        ws = ops.workspace(nb, self.dev, "lin_wgrad")
        call("tacorl_linear_wgrad", n, ops.ptr_array(xs), ldx, ops.ptr_array(dzs), ld_dz, ops.int_array([M] * n), K, O,
             ops.ptr_array(dws), ops.ptr_array(dbs), 0, compute, ptr(ws), ws.numel(), ops.stream())

    def _dgrad(self, dz, ld_dz, w, out, ld_out, M, O, I, compute, src=None, ld_src=0, act=ACT_NONE, addend=None,
               ld_add=0):
        nb = ops.L.lib().tacorl_linear_dgrad_ws_bytes(1, ops.int_array([M]), O, I)  # split reduction for skinny outputs
        ws = ops.workspace(nb, self.dev, "lin_dgrad")
        call("tacorl_linear_dgrad_splitk", 1, ops.ptr_array([dz]), ld_dz, ops.ptr_array([w]), ops.ptr_array([out]), ld_out,
             ops.ptr_array([src]) if src is not None else None, ld_src, act,
             ops.ptr_array([addend]) if addend is not None else None, ld_add, ops.int_array([M]), O, I, compute,
             ptr(ws), ws.numel(), ops.stream())

    def _ln_bwd(self, dy, x, res, w, stats, dv, dw, db, R, D):
        nb = ops.L.lib().tacorl_add_layernorm_bwd_ws_bytes(R, D)
        ws = ops.workspace(nb, self.dev, "ln_bwd")
        call("tacorl_add_layernorm_bwd", ptr(dy), ptr(x), ptr(res), w, ptr(stats), ptr(dv), dw, db, R, D, 0, ptr(ws),
             ws.numel(), ops.stream())

    def prepare_backward(self, B):
        """Weights-only preparation of the fused backward: linear1.weight^T / linear2.weight^T of every layer as bf16."""
        blk, D, FF = self.blk, self.D, self.FF
        if getattr(self, "_wt", None) is None or self._lnpart.numel() != self.L * 2 * B * 64:
            ops.note_alloc()
            bf = lambda n: torch.zeros(n, device=self.dev, dtype=torch.bfloat16)  # noqa: E731
            self._wt = [t for _ in range(self.L) for t in (bf(FF * D), bf(FF * D), bf(D * D), bf(3 * D * D))]
            self._lnpart = torch.zeros(self.L * 2 * B * 64, device=self.dev)
        srcs, shp = [], []
        for l in range(self.L):
            p = f"transformer_encoder.layers.{l}."
            srcs += [blk.p(p + "linear1.weight"), blk.p(p + "linear2.weight"), blk.p(p + "self_attn.out_proj.weight"), blk.p(p + "self_attn.in_proj_weight")]
            shp += [(FF, D), (D, FF), (D, D), (3 * D, D)]
        for i in range(0, len(srcs), 16):  # one launch for all of them (round 5: 4 L launches before)
            j = slice(i, i + 16)
            call("tacorl_transpose_to_bf16_batch", len(srcs[j]), ops.ptr_array(srcs[j]), ops.ptr_array(self._wt[j]),
                 ops.int_array([r for r, _ in shp[j]]), ops.int_array([c for _, c in shp[j]]), ops.stream())

    def backward(self, d_head, B, T, compute, wgrad_stream=None, prepared=False):
        """d_head: (B, 2A) gradient w.r.t. [mean | var_raw].  Fills self.blk.grad and returns the
        (B*T, D) gradient w.r.t. the (padded) input embeddings.
        wgrad_stream: the Linear weight gradients - read only by the optimiser - are issued on that stream (a branch
        of a captured graph) beside the dependent input-gradient chain; the caller joins it.  The gradient buffers
        those launches read are per layer, so the chain never overwrites what a pending weight gradient still needs."""
        blk, D, R, FF, FC, A2 = self.blk, self.D, B * T, self.FF, self.FC, 2 * self.A
        if getattr(self, "_bshape", None) != (B, T):
            ops.note_alloc()
            f = lambda *s: torch.zeros(*s, device=self.dev)  # noqa: E731
            per = lambda *s: [f(*s) for _ in range(self.L)]  # noqa: E731
            self.d_fc, self.d_pool = f(B, FC), f(B, D)
            self.dx, self.d_x1, self.d_att = f(R, D), f(R, D), f(R, D)
            self.dv, self.dv1, self.d_ff1, self.d_qkv = per(R, D), per(R, D), per(R, FF), per(R, 3 * D)
            self.dvb, self.dv1b = per(R, D), per(R, D)  # branch gradients behind dropout2 / dropout1 (train mode)
            self._bshape = (B, T)

        def side(fn):
            if wgrad_stream is None:
                return fn()
            wgrad_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(wgrad_stream):
                fn()

        fused = (getattr(self, "_fused_saved", None) == (B, T) and getattr(self, "fused_backward", True)
                 and not getattr(self, "_dropping", False) and T == 16)  # (the one-launch backward exists for window 16)
        wt_ready = None
        if fused and not prepared:
            # W1^T / W2^T as bf16 for the fused chain: weights only, so on the weight-gradient stream.  NB that stream's queue
            # may hold another network's weight gradients (PlayLMP: the action decoder's), which the chain below then waits
            # for.  Measured, that wait PAYS: with the transposes issued early elsewhere (prepare_backward(), prepared=True)
            # the decoder's weight gradients run beside this chain and the encoder backward and the step is 20-40 us
            # slower (chip-wide kernels beside a latency-bound chain).
            side(lambda: self.prepare_backward(B))
            if wgrad_stream is not None:
                wt_ready = torch.cuda.Event()
                wt_ready.record(wgrad_stream)
        composed = fused and getattr(self, "_composed", False)
        if composed:
            def head_grads():  # fc_out and d_fc exist only for the two weight gradients: all of it beside the chain
                self._lin(self.pooled, D, blk.p("fc.weight"), blk.p("fc.bias"), self.fc_out, B, D, FC, ACT_NONE, compute)
                self._wgrad(self.fc_out, FC, d_head, A2, B, FC, A2, blk.g("mean_fc.weight"), blk.g("mean_fc.bias"), compute)
                self._dgrad(d_head, A2, blk.p("mean_fc.weight"), self.d_fc, FC, B, A2, FC, compute)
                self._wgrad(self.pooled, D, self.d_fc, FC, B, D, FC, blk.g("fc.weight"), blk.g("fc.bias"), compute)
            side(head_grads)
        else:
            if getattr(self, "_composed", False):
                # (the forward took the head as one composed affine map - fc_out was never formed - but this backward is the
                # per-op chain, e.g. window 32: form it now, mean_fc's weight gradient reads it)
                self._lin(self.pooled, D, blk.p("fc.weight"), blk.p("fc.bias"), self.fc_out, B, D, FC, ACT_NONE, compute)
            side(lambda: self._wgrad(self.fc_out, FC, d_head, A2, B, FC, A2, blk.g("mean_fc.weight"), blk.g("mean_fc.bias"), compute))
            self._dgrad(d_head, A2, blk.p("mean_fc.weight"), self.d_fc, FC, B, A2, FC, compute)
            side(lambda: self._wgrad(self.pooled, D, self.d_fc, FC, B, D, FC, blk.g("fc.weight"), blk.g("fc.bias"), compute))
            self._dgrad(self.d_fc, FC, blk.p("fc.weight"), self.d_pool, D, B, FC, D, compute)
        if fused:
            # the whole input-gradient chain of the encoder layers in one launch (csrc/pr_fused.hip); the weight gradients
            # stay per-op GEMMs on the side stream and read the dZ operands that launch wrote
            if wt_ready is not None:
                torch.cuda.current_stream().wait_event(wt_ready)
            saved, dz, lng = [], [], []
            for l in range(self.L):
                p = f"transformer_encoder.layers.{l}."
                saved += [self.x[2 * l], self.qkv[l], self.att[l], self.proj[l], self.x[2 * l + 1], self.ff1[l], self.ff2[l],
                          self.stats[2 * l], self.stats[2 * l + 1]]
                dz += [self.dv[l], self.d_ff1[l], self.dv1[l], self.d_qkv[l]]
                lng += [blk.g(p + "norm1.weight"), blk.g(p + "norm1.bias"), blk.g(p + "norm2.weight"), blk.g(p + "norm2.bias")]
            call("tacorl_pr_encoder_bwd_fused", ptr(blk.param), self._foff, None if composed else ptr(self.d_pool), ptr(d_head),
                 ptr(self._Wc), A2, ptr(self.dx), ops.ptr_array(saved), ops.ptr_array(dz), ops.ptr_array(self._wt if getattr(self, "bwd_transposed_proj", True) else
                                                                              [t if k % 4 < 2 else None for k, t in enumerate(self._wt)]),
                 ptr(self._lnpart), ops.ptr_array(lng),
                 B, D, T, self.H, FF, self.L, ops.stream())

            def wgrads():
                # (round 5: the layers' gradients of one shape share a launch - 4 GEMM + 4 slab-reduce launches instead of
                # 4 L of each; every one of them is launch / latency bound.  pr_wgrad_batched = False: layer by layer)
                lp = [f"transformer_encoder.layers.{l}." for l in range(self.L)]
                ls = range(self.L)
                if getattr(self, "pr_wgrad_batched", True) and self.L <= 16:
                    self._wgrad_n([self.ff1[l] for l in ls], FF, [self.dv[l] for l in ls], D, R, FF, D,
                                  [blk.g(p + "linear2.weight") for p in lp], [blk.g(p + "linear2.bias") for p in lp], compute)
                    self._wgrad_n([self.x[2 * l + 1] for l in ls], D, [self.d_ff1[l] for l in ls], FF, R, D, FF,
                                  [blk.g(p + "linear1.weight") for p in lp], [blk.g(p + "linear1.bias") for p in lp], compute)
                    self._wgrad_n([self.att[l] for l in ls], D, [self.dv1[l] for l in ls], D, R, D, D,
                                  [blk.g(p + "self_attn.out_proj.weight") for p in lp], [blk.g(p + "self_attn.out_proj.bias") for p in lp], compute)
                    self._wgrad_n([self.x[2 * l] for l in ls], D, [self.d_qkv[l] for l in ls], 3 * D, R, D, 3 * D,
                                  [blk.g(p + "self_attn.in_proj_weight") for p in lp], [blk.g(p + "self_attn.in_proj_bias") for p in lp], compute)
                    return
                for l in reversed(range(self.L)):
                    p = f"transformer_encoder.layers.{l}."
                    self._wgrad(self.ff1[l], FF, self.dv[l], D, R, FF, D, blk.g(p + "linear2.weight"), blk.g(p + "linear2.bias"), compute)
                    self._wgrad(self.x[2 * l + 1], D, self.d_ff1[l], FF, R, D, FF, blk.g(p + "linear1.weight"), blk.g(p + "linear1.bias"), compute)
                    self._wgrad(self.att[l], D, self.dv1[l], D, R, D, D, blk.g(p + "self_attn.out_proj.weight"),
                                blk.g(p + "self_attn.out_proj.bias"), compute)
                    self._wgrad(self.x[2 * l], D, self.d_qkv[l], 3 * D, R, D, 3 * D, blk.g(p + "self_attn.in_proj_weight"),
                                blk.g(p + "self_attn.in_proj_bias"), compute)
            side(wgrads)
            call("tacorl_reduce_rows_mod", ptr(self.dx), D, blk.g("position_embeddings.weight"), D, T, D, B, ops.stream())
            if self.T_max > T:
                blk.grad_views["position_embeddings.weight"][T:].zero_()
            return self.dx
        call("tacorl_bcast_over_t", ptr(self.d_pool), ptr(self.dx), B, T, D, 1.0 / T, 0, ops.stream())
        # train mode: the forward dropped ff2 / ff1 / proj / attention probabilities / embeddings in place (the saved
        # buffers ARE the dropped values, so the weight gradients and the fused ReLU mask see what the next layer saw);
        # a branch's gradient takes the same keep mask and scale, the residual path takes the unmasked one
        drop = getattr(self, "_dropping", False)
        ks = 1.0 / (1.0 - self.dropout_p) if drop else 1.0
        for l in reversed(range(self.L)):
            p = f"transformer_encoder.layers.{l}."
            xin, x1 = self.x[2 * l], self.x[2 * l + 1]
            dv, dv1, d_ff1, d_qkv = self.dv[l], self.dv1[l], self.d_ff1[l], self.d_qkv[l]
            self._ln_bwd(self.dx, x1, self.ff2[l], blk.p(p + "norm2.weight"), self.stats[2 * l + 1], dv,
                         blk.g(p + "norm2.weight"), blk.g(p + "norm2.bias"), R, D)
            dvb = dv
            if drop:
                dvb = self.dvb[l]
                ops.copy_cols(dv, 0, D, dvb, 0, D, R, D)
                self._drop(dvb, 4 + 4 * l, R * D)
            side(lambda: self._wgrad(self.ff1[l], FF, dvb, D, R, FF, D, blk.g(p + "linear2.weight"), blk.g(p + "linear2.bias"),
                                     compute))
            self._dgrad(dvb, D, blk.p(p + "linear2.weight"), d_ff1, FF, R, D, FF, compute, src=self.ff1[l],
                        ld_src=FF, act=ACT_RELU)
            if drop:  # (the ReLU mask taken from the dropped ff1 already zeroed the dropped units: this adds the scale)
                self._drop(d_ff1, 3 + 4 * l, R * FF)
            side(lambda: self._wgrad(x1, D, d_ff1, FF, R, D, FF, blk.g(p + "linear1.weight"), blk.g(p + "linear1.bias"), compute))
            self._dgrad(d_ff1, FF, blk.p(p + "linear1.weight"), self.d_x1, D, R, FF, D, compute, addend=dv, ld_add=D)
            self._ln_bwd(self.d_x1, xin, self.proj[l], blk.p(p + "norm1.weight"), self.stats[2 * l], dv1,
                         blk.g(p + "norm1.weight"), blk.g(p + "norm1.bias"), R, D)
            dv1b = dv1
            if drop:
                dv1b = self.dv1b[l]
                ops.copy_cols(dv1, 0, D, dv1b, 0, D, R, D)
                self._drop(dv1b, 2 + 4 * l, R * D)
            side(lambda: self._wgrad(self.att[l], D, dv1b, D, R, D, D, blk.g(p + "self_attn.out_proj.weight"),
                                     blk.g(p + "self_attn.out_proj.bias"), compute))
            self._dgrad(dv1b, D, blk.p(p + "self_attn.out_proj.weight"), self.d_att, D, R, D, D, compute)
            if drop:
                call("tacorl_attention_dropout_bwd", ptr(self.qkv[l]), ptr(self.d_att), ptr(d_qkv),
                     ptr(self.keep[1 + 4 * l]), ks, B, T, D, self.H, ops.stream())
            else:
                call("tacorl_attention_bwd", ptr(self.qkv[l]), ptr(self.d_att), ptr(d_qkv), B, T, D, self.H, ops.stream())
            side(lambda: self._wgrad(xin, D, d_qkv, 3 * D, R, D, 3 * D, blk.g(p + "self_attn.in_proj_weight"),
                                     blk.g(p + "self_attn.in_proj_bias"), compute))
            self._dgrad(d_qkv, 3 * D, blk.p(p + "self_attn.in_proj_weight"), self.dx, D, R, 3 * D, D, compute,
                        addend=dv1, ld_add=D)
        if drop:
            self._drop(self.dx, 0, R * D)
        # position embeddings: sum over the batch of rows with the same t
        call("tacorl_reduce_rows_mod", ptr(self.dx), D, blk.g("position_embeddings.weight"), D, T, D, B, ops.stream())
        if self.T_max > T:
            blk.grad_views["position_embeddings.weight"][T:].zero_()
        return self.dx
