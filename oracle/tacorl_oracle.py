"""CPU oracle for the TACO-RL offline training step.  TEST INFRASTRUCTURE ONLY.

A functional, noise-injected, fp32 restatement (plain torch-CPU ops + autograd)
of the reference hot path.  It is pinned against
tests/golden/*.npz, which oracle/gen_golden.py produced by running the unmodified
reference (tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file; the product package
(tacorl_amd/) never does.

Conventions
* ``P`` is a dict {state-dict name: tensor} with the reference's key layout
  (SURVEY 8a note 9); conv weights are logical OIHW.
* every random draw is an explicit argument (``noise`` dict); order and shapes
  follow the reference's draw order (SURVEY 8a note 1).
* ``faithful=True`` re-encodes images exactly as often as the reference does
  ((24+12n)B + 16B encoder images per TACORL step) - used for the CPU-baseline
  timing; ``faithful=False`` encodes each unique (encoder, image set) once.
  Both give the same numbers.
* ``operand_rounding(torch.bfloat16)`` (context manager) evaluates the SAME algorithm with the operand
  rounding of the GPU's bf16 MFMA mode: both operands of every Linear / Conv2d contraction - forward,
  input-gradient and weight-gradient - are rounded to bf16, accumulation and everything else stay fp32
  (master weights, activations, reductions, transcendental functions).  That is the yardstick for the bf16
  kernels (tests hold them to it far tighter than to the fp32 reference); without the context the oracle is
  the exact fp32 restatement that the goldens pin.
"""
import math

import torch
import torch.nn.functional as F

LOG_SIG_MAX, LOG_SIG_MIN = 2.0, -5.0  # reference networks/actor_critic/actor.py:12-15
MEAN_MIN, MEAN_MAX = -9.0, 9.0
LOG2 = math.log(2.0)


# ------------------------------------------------------------- contractions (exact / bf16 operands)
_OPERAND_DTYPE = None
_ROUND_ONLY = None   # None: every contraction; else a set of contraction classes ("conv", "rnn", "attn", "linear")
_REGION = []         # class of the contractions being evaluated (innermost first wins), "linear" outside any region


class operand_rounding:
    """with operand_rounding(torch.bfloat16): ... - see the module docstring.
    only={"conv", "rnn", "attn", "linear"} (any subset) restricts the rounding to those contraction classes - the
    convolutions, the action decoder's recurrent network and heads, the plan-recognition transformer, every other Linear -
    for bisecting which rounding a gradient is sensitive to (profiles/r05_playlmp_bf16_bisect.md); the backward of a
    contraction rounds exactly when its forward did."""

    def __init__(self, dtype, only=None):
        self.dtype, self.only = dtype, (None if only is None else frozenset(only))

    def __enter__(self):
        global _OPERAND_DTYPE, _ROUND_ONLY
        self.prev, _OPERAND_DTYPE = (_OPERAND_DTYPE, _ROUND_ONLY), self.dtype
        _ROUND_ONLY = self.only

    def __exit__(self, *exc):
        global _OPERAND_DTYPE, _ROUND_ONLY
        _OPERAND_DTYPE, _ROUND_ONLY = self.prev
        return False


class _region:
    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        _REGION.append(self.kind)

    def __exit__(self, *exc):
        _REGION.pop()
        return False


def _rounds(kind):
    if _OPERAND_DTYPE is None:
        return False
    return _ROUND_ONLY is None or (_REGION[-1] if _REGION and kind == "linear" else kind) in _ROUND_ONLY


def _r(t, dt=None):
    return t.to(dt if dt is not None else _OPERAND_DTYPE).to(torch.float32)


class _RoundedLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = _r(x), _r(w)
        ctx.save_for_backward(xr, wr)
        ctx.has_b, ctx.dt = b is not None, _OPERAND_DTYPE
        return F.linear(xr, wr, b)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = _r(dy, ctx.dt)
        d2, x2 = dyr.reshape(-1, dyr.shape[-1]), xr.reshape(-1, xr.shape[-1])
        return dyr @ wr, d2.t() @ x2, (d2.sum(0) if ctx.has_b else None)


class _RoundedConv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride):
        xr, wr = _r(x), _r(w)
        ctx.save_for_backward(xr, wr)
        ctx.stride, ctx.has_b, ctx.dt = stride, b is not None, _OPERAND_DTYPE
        return F.conv2d(xr, wr, b, stride=stride)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = _r(dy, ctx.dt)
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, stride=ctx.stride)
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, stride=ctx.stride)
        return dx, dw, (dyr.sum(dim=(0, 2, 3)) if ctx.has_b else None), None


def _linear(x, w, b=None):
    return _RoundedLinear.apply(x, w, b) if _rounds("linear") else F.linear(x, w, b)


def _conv2d(x, w, b=None, stride=1):
    return _RoundedConv2d.apply(x, w, b, stride) if _rounds("conv") else F.conv2d(x, w, b, stride=stride)


# ----------------------------------------------------------------------------- A1/A2
def spatial_softargmax(x, temperature):
    """reference networks/visual_encoders/utils.py:39-76 (normalize=False).
    x (N,C,h,w) -> (N,2C) interleaved [x0,y0,x1,y1,...] in pixel units."""
    n, c, h, w = x.shape
    sm = F.softmax(x.reshape(n * c, h * w) / temperature, dim=1).reshape(n, c, h, w)
    xs = torch.arange(w, dtype=x.dtype)
    ys = torch.arange(h, dtype=x.dtype)
    ex = (sm * xs.view(1, 1, 1, w)).sum(dim=(2, 3))
    ey = (sm * ys.view(1, 1, h, 1)).sum(dim=(2, 3))
    return torch.stack([ex, ey], dim=-1).reshape(n, 2 * c)


def encoder_fwd(P, pre, img):
    """LMPVisionEncoder.forward, reference networks/visual_encoders/encoder.py:369-419.
    img (N,3,H,W) -> (N,32)."""
    x = F.relu(_conv2d(img, P[pre + "model.0.weight"], P[pre + "model.0.bias"], stride=4))
    x = F.relu(_conv2d(x, P[pre + "model.2.weight"], P[pre + "model.2.bias"], stride=2))
    x = F.relu(_conv2d(x, P[pre + "model.4.weight"], P[pre + "model.4.bias"], stride=1))
    x = spatial_softargmax(x, P[pre + "model.6.temperature"])
    x = F.relu(_linear(x, P[pre + "fc_layers.0.weight"], P[pre + "fc_layers.0.bias"]))
    return _linear(x, P[pre + "fc_layers.3.weight"], P[pre + "fc_layers.3.bias"])


def late_fusion(P, pre, obs, cams):
    """LateFusion.get_state_from_observation (representation_network.py:36-65):
    concat per-modality embeddings in ``cams`` order."""
    return torch.cat([encoder_fwd(P, f"{pre}networks.{c}.", obs[c]) for c in cams], dim=-1)


# -------------------------------------------------------------------------------- A3
def goal_encoder(P, pre, x):
    """VisualGoalEncoder.forward, reference visual_encoders/goal_encoder.py:17-33."""
    x = F.relu(_linear(x, P[pre + "mlp.0.weight"], P[pre + "mlp.0.bias"]))
    x = F.relu(_linear(x, P[pre + "mlp.2.weight"], P[pre + "mlp.2.bias"]))
    return _linear(x, P[pre + "mlp.4.weight"], P[pre + "mlp.4.bias"])


# -------------------------------------------------------------------------------- A6
def policy(P, pre, s, n_layers=3, discrete_gripper=False):
    """MLPPolicy.forward, reference actor_critic/actor.py:252-270."""
    x = s
    for i in range(n_layers):
        x = F.silu(_linear(x, P[f"{pre}fc_layers.{i}.weight"], P[f"{pre}fc_layers.{i}.bias"]))
    mean = torch.clamp(_linear(x, P[pre + "fc_mean.weight"], P[pre + "fc_mean.bias"]), MEAN_MIN, MEAN_MAX)
    log_std = torch.clamp(_linear(x, P[pre + "fc_log_std.weight"], P[pre + "fc_log_std.bias"]),
                          LOG_SIG_MIN, LOG_SIG_MAX)
    if discrete_gripper:
        logits = _linear(x, P[pre + "gripper_action.weight"], P[pre + "gripper_action.bias"])
        return mean, log_std.exp(), logits
    return mean, log_std.exp()


# -------------------------------------------------------------------------------- A8
def qnet(P, pre, s, a, n_layers=3):
    """Critic.forward + MLPQNetwork.forward, reference actor_critic/critic.py:24-30,92-97."""
    x = torch.cat([s, a], dim=-1)
    for i in range(n_layers):
        x = F.silu(_linear(x, P[f"{pre}fc_layers.{i}.weight"], P[f"{pre}fc_layers.{i}.bias"]))
    return _linear(x, P[pre + "out.weight"], P[pre + "out.bias"])


# -------------------------------------------------------------------------------- A7
def normal_logprob(z, mu, std):
    """Independent(Normal).log_prob summed over the last dim (torch semantics)."""
    var = std * std
    return (-((z - mu) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(-1)


def tanh_logprob(z, mu, std):
    """TanhNormal._log_prob_from_pre_tanh, reference utils/distributions.py:86-96 -> (...,1)."""
    corr = -2.0 * (LOG2 - z - F.softplus(-2.0 * z)).sum(dim=-1)
    return (normal_logprob(z, mu, std) + corr).unsqueeze(-1)


def ref_atanh(x):
    """reference utils/misc.py:297-300."""
    return 0.5 * torch.log((1 + x).clamp(min=1e-6) / (1 - x).clamp(min=1e-6))


def tanh_logprob_of_value(value, mu, std):
    """TanhNormal.log_prob(value), distributions.py:98-109 (clamp +-0.999, atanh)."""
    return tanh_logprob(ref_atanh(torch.clamp(value, -0.999, 0.999)), mu, std)


def gumbel_argmax(norm_logits, u01):
    """GumbelSoftmax.sample, distributions.py:28-38 (logits already normalised)."""
    return torch.argmax(norm_logits - torch.log(-torch.log(u01)), dim=-1)


def gumbel_rsample_hard_index(norm_logits, rand01, temperature=0.5):
    """argmax of RelaxedOneHotCategorical.rsample (torch ExpRelaxedCategorical.rsample);
    reference actor.py:87-88.  Only the index survives (no gradient)."""
    eps = torch.finfo(rand01.dtype).eps
    u = rand01.clamp(min=eps, max=1 - eps)
    g = -((-(u.log())).log())
    scores = (norm_logits + g) / temperature
    return torch.argmax(scores - scores.logsumexp(dim=-1, keepdim=True), dim=-1)


def gripper_logprob(logits, idx):
    """GumbelSoftmax.log_prob, distributions.py:50-58 -> (...,1)."""
    norm = logits - logits.logsumexp(dim=-1, keepdim=True)
    lsm = F.log_softmax(norm, dim=-1)
    return torch.gather(lsm, -1, idx.long().unsqueeze(-1))


# ------------------------------------------------------------------------------- A11
def _layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def plan_recognition(P, pre, emb, n_heads=8, n_layers=2, min_std=1e-4, dropout=None):
    """PlanRecognitionTransformersNetwork.forward,
    reference plan_encoders/plan_recognition_transformer.py:70-105 with torch's
    nn.TransformerEncoderLayer defaults (post-norm, ReLU).  emb (B,T,D) -> mu,std (B,A).
    dropout = (p, masks): train mode; masks = the 1 + 4 * n_layers keep masks in the reference's draw order and
    layouts - embedding (T,B,D) (:87), then per layer attention probabilities (B,H,T,T), dropout1 (T,B,D), FFN
    dropout (T,B,FF), dropout2 (T,B,D) (torch nn.TransformerEncoderLayer._sa_block / _ff_block)."""
    p_drop, masks = dropout if dropout is not None else (0.0, None)
    ks = 1.0 / (1.0 - p_drop)

    def drop_tbd(x, m):  # x batch-major (B,T,*), mask sequence-major (T,B,*)
        return x * m.to(x.dtype).permute(1, 0, 2) * ks

    B, T, D = emb.shape
    pad = (-D) % n_heads
    if pad:
        emb = torch.cat([emb, torch.zeros(B, T, pad, dtype=emb.dtype)], dim=-1)
        D += pad
    x = emb + P[pre + "position_embeddings.weight"][:T].unsqueeze(0)
    if masks is not None:
        x = drop_tbd(x, masks[0])
    hd = D // n_heads
    _REGION.append("attn")  # (contraction class for operand_rounding(only=...); popped before returning)
    for l in range(n_layers):
        lp = f"{pre}transformer_encoder.layers.{l}."
        mk = masks[1 + 4 * l: 5 + 4 * l] if masks is not None else None
        qkv = _linear(x, P[lp + "self_attn.in_proj_weight"], P[lp + "self_attn.in_proj_bias"])
        q, k, v = qkv.split(D, dim=-1)
        sh = lambda t: t.reshape(B, T, n_heads, hd).permute(0, 2, 1, 3)  # noqa: E731
        q, k, v = sh(q), sh(k), sh(v)
        att = torch.softmax((q / math.sqrt(hd)) @ k.transpose(-1, -2), dim=-1)
        if mk is not None:
            att = att * mk[0].to(att.dtype) * ks
        o = (att @ v).permute(0, 2, 1, 3).reshape(B, T, D)
        o = _linear(o, P[lp + "self_attn.out_proj.weight"], P[lp + "self_attn.out_proj.bias"])
        if mk is not None:
            o = drop_tbd(o, mk[1])
        x = _layer_norm(x + o, P[lp + "norm1.weight"], P[lp + "norm1.bias"])
        hdn = F.relu(_linear(x, P[lp + "linear1.weight"], P[lp + "linear1.bias"]))
        if mk is not None:
            hdn = drop_tbd(hdn, mk[2])
        f = _linear(hdn, P[lp + "linear2.weight"], P[lp + "linear2.bias"])
        if mk is not None:
            f = drop_tbd(f, mk[3])
        x = _layer_norm(x + f, P[lp + "norm2.weight"], P[lp + "norm2.bias"])
    x = _linear(x, P[pre + "fc.weight"], P[pre + "fc.bias"]).mean(dim=1)
    mean = _linear(x, P[pre + "mean_fc.weight"], P[pre + "mean_fc.bias"])
    std = F.softplus(_linear(x, P[pre + "variance_fc.weight"], P[pre + "variance_fc.bias"])) + min_std
    _REGION.pop()
    return mean, std


# ------------------------------------------------------------------------------- A12
def action_decoder_fwd(P, pre, plan, emb, n_mix=10, n_layers=2, h0=None, return_hidden=False):
    """ActionDecoderLogistic.forward (rnn_decoder = 2-layer ReLU nn.RNN, batch_first),
    reference action_decoders/action_decoder_logistic.py:268-300, rnn_models.py:5-16.
    h0 (n_layers,B,H): initial hidden state (the `act` path, :87-97); return_hidden appends h_n."""
    B, T, _ = emb.shape
    x = torch.cat([plan.unsqueeze(1).expand(-1, T, -1), emb], dim=-1)
    hn = []
    _REGION.append("rnn")  # (contraction class of everything below, for operand_rounding(only=...); popped before returning)
    for l in range(n_layers):
        wi, wh = P[f"{pre}rnn.weight_ih_l{l}"], P[f"{pre}rnn.weight_hh_l{l}"]
        bi, bh = P[f"{pre}rnn.bias_ih_l{l}"], P[f"{pre}rnn.bias_hh_l{l}"]
        h = torch.zeros(B, wh.shape[0], dtype=emb.dtype) if h0 is None else h0[l]
        xin = _linear(x, wi, bi)
        outs = []
        for t in range(T):
            h = F.relu(xin[:, t] + _linear(h, wh, bh))
            outs.append(h)
        hn.append(h)
        x = torch.stack(outs, dim=1)
    probs = _linear(x, P[pre + "prob_fc.weight"], P[pre + "prob_fc.bias"])
    means = _linear(x, P[pre + "mean_fc.weight"], P[pre + "mean_fc.bias"])
    log_scales = torch.clamp(_linear(x, P[pre + "log_scale_fc.weight"], P[pre + "log_scale_fc.bias"]),
                             min=LOG_SIG_MIN)
    grip = _linear(x, P[pre + "gripper_fc.weight"], P[pre + "gripper_fc.bias"])
    _REGION.pop()
    v = lambda t: t.reshape(B, T, -1, n_mix)  # noqa: E731
    if return_hidden:
        return v(probs), v(log_scales), v(means), grip, torch.stack(hn)
    return v(probs), v(log_scales), v(means), grip


def action_decoder_act(P, pre, plan, emb, h0, rand_a, rand_b):
    """ActionDecoderLogistic.act (:87-97): one decoder step, hidden state in / out, sampled action (B,1,7)."""
    lp, ls, mm, gr, hn = action_decoder_fwd(P, pre, plan, emb, h0=h0, return_hidden=True)
    return logistic_sample(lp, ls, mm, gr, rand_a, rand_b), hn


def actor_get_actions(P, net, obs, goal, spec, deterministic=False, reparameterize=False, noise=None):
    """VisualActorWrapper.get_actions -> Actor.get_actions (visual_actor_wrapper.py:64-66, actor.py:65-111).
    noise: 'eps' (B,Ac) and, for the discrete gripper, 'gumbel_u' (B,2) U(0,1)."""
    s = _emb(P, net, obs, goal, spec)
    po = policy(P, net + "actor.policy.", s, discrete_gripper=spec.discrete_gripper)
    mu, std = po[0], po[1]
    if deterministic:
        a = torch.tanh(mu)
        if spec.discrete_gripper:
            gi = torch.argmax(torch.softmax(po[2], dim=-1), dim=-1)
            a = torch.cat([a, gi.unsqueeze(-1) * 2.0 - 1], dim=-1)
        return a, torch.zeros_like(a)
    z = mu + noise["eps"] * std
    a, log_pi = torch.tanh(z), tanh_logprob(z, mu, std)
    if spec.discrete_gripper:
        nl = po[2] - po[2].logsumexp(dim=-1, keepdim=True)
        gi = gumbel_rsample_hard_index(nl, noise["gumbel_u"]) if reparameterize else gumbel_argmax(nl, noise["gumbel_u"])
        log_pi = log_pi + gripper_logprob(po[2], gi)
        a = torch.cat([a, gi.unsqueeze(-1) * 2.0 - 1], dim=-1)
    return a, log_pi


def logistic_mixture_loss(logit_probs, log_scales, means, grip, actions, num_classes=10,
                          gripper_alpha=1.0):
    """ActionDecoderLogistic._loss/_logistic_loss, action_decoder_logistic.py:110-235
    (bounds +-1, discrete gripper)."""
    a = actions[:, :, :-1].unsqueeze(-1)
    log_scales = torch.clamp(log_scales, min=LOG_SIG_MIN)
    centered = a - means
    inv_std = torch.exp(-log_scales)
    half_bin = 1.0 / (num_classes - 1)  # act_range(=1) / (num_classes-1)
    plus_in = inv_std * (centered + half_bin)
    min_in = inv_std * (centered - half_bin)
    cdf_plus, cdf_min = torch.sigmoid(plus_in), torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    mid_in = inv_std * centered
    log_pdf_mid = mid_in - log_scales - 2.0 * F.softplus(mid_in)
    cdf_delta = cdf_plus - cdf_min
    a_b = a.expand_as(means)
    log_probs = torch.where(
        a_b < -1.0 + 1e-3, log_cdf_plus,
        torch.where(a_b > 1.0 - 1e-3, log_one_minus_cdf_min,
                    torch.where(cdf_delta > 1e-5, torch.log(torch.clamp(cdf_delta, min=1e-12)),
                                log_pdf_mid - math.log((num_classes - 1) / 2))))
    log_probs = log_probs + F.log_softmax(logit_probs, dim=-1)
    m = log_probs.max(dim=-1, keepdim=True).values  # utils/misc.py:289-294
    lse = m.squeeze(-1) + torch.log(torch.exp(log_probs - m).sum(dim=-1))
    logistic = -lse.sum(dim=-1).mean()
    gt = actions[:, :, -1].clone()
    gt[gt == -1] = 0
    ce = F.cross_entropy(grip.reshape(-1, 2), gt.reshape(-1).long())
    return logistic + gripper_alpha * ce


def logistic_sample(logit_probs, log_scales, means, grip, rand_a, rand_b):
    """ActionDecoderLogistic._sample, action_decoder_logistic.py:238-266."""
    r1, r2 = 1e-5, 1.0 - 1e-5
    t = logit_probs - torch.log(-torch.log((r1 - r2) * rand_a + r2))
    dist = F.one_hot(torch.argmax(t, -1), logit_probs.shape[-1]).to(means.dtype)
    ls = (dist * log_scales).sum(-1)
    mu = (dist * means).sum(-1)
    u = (r1 - r2) * rand_b + r2
    act = mu + torch.exp(ls) * (torch.log(u) - torch.log(1.0 - u))
    g = torch.tensor([-1.0, 1.0])[grip.argmax(dim=-1)]
    return torch.cat([act, g.unsqueeze(-1)], 2)


# -------------------------------------------------------------------------------- K8
class Adam:
    """torch.optim.Adam semantics (betas .9/.999, eps 1e-8), restated."""

    def __init__(self, names, lr):
        self.names, self.lr, self.t = list(names), lr, 0
        self.m, self.v = {}, {}

    def step(self, P, grads):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        with torch.no_grad():
            for n in self.names:
                g = grads[n]
                if n not in self.m:
                    self.m[n], self.v[n] = torch.zeros_like(g), torch.zeros_like(g)
                self.m[n].mul_(b1).add_(g, alpha=1 - b1)
                self.v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(eps)
                P[n].addcdiv_(self.m[n], denom, value=-self.lr / bc1)


def clip_grads_(grads, names, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2, eps 1e-6) on the named grads, in place."""
    total = torch.sqrt(sum((grads[n].double() ** 2).sum() for n in names)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for n in names:
        grads[n] = grads[n] * coef
    return total


def _grads(loss, P, names, retain=True):
    gs = torch.autograd.grad(loss, [P[n] for n in names], retain_graph=retain, allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, gs)}


def group_names(P, prefix):
    return [n for n in P if n.startswith(prefix)]


# ---------------------------------------------------------------------------- A5, A9
class ACSpec:
    """Static description of an actor-critic step (what the reference reads from cfg)."""

    def __init__(self, cams, goal_cams, action_dim, n=4, discount=0.99, tau=0.005, actor_lr=1e-4,
                 critic_lr=3e-4, deterministic_backup=False, reward_scale=1.0, bc_epochs=0,
                 clip_grad_val=1.0, conservative_weight=1.0, lagrange_thresh=5.0, temp=1.0,
                 with_lagrange=True, discrete_gripper=False, target_entropy=-7.0,
                 finetune_action_decoder=False, action_decoder_lr=1e-4, ac_cams=None, pr_cams=None):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def _emb(P, net, obs, goal, spec):
    """Visual*Wrapper.get_emb_representation (visual_actor_wrapper.py:41-62):
    [enc(obs) || goal_enc(enc(goal))]."""
    e = late_fusion(P, net + "encoder.", obs, spec.cams)
    g = goal_encoder(P, net + "goal_encoder.", late_fusion(P, net + "encoder.", goal, spec.goal_cams))
    return torch.cat([e, g], dim=-1)


def _expand(x, n):
    """utils/misc.py:132-137 expand_array: index k*B+b."""
    return x.unsqueeze(0).expand(n, *x.shape).reshape(-1, *x.shape[1:])


def ac_update(P, opts, spec, obs, goal, next_obs, action, rewards, dones, noise, epoch, logs,
              faithful=False):
    """CQL_Offline.compute_update (cql_offline_lightning.py:470-542) with injected noise.

    noise keys: 'eps_pi' (B,Ac) actor rsample, 'eps_next' (B,Ac), 'u_rand' (nB,A) U(0,1),
    'eps_cur' (n,B,Ac), 'eps_nxt' (n,B,Ac); discrete gripper adds 'g_pi' (B,2) rand,
    'g_next' (B,2), 'g_cur' (n,B,2), 'g_nxt' (n,B,2) U(0,1).
    rewards/dones: (B,1).  Mutates P (optimiser steps, Polyak) and returns grads per group."""
    n, B, dg = spec.n, action.shape[0], spec.discrete_gripper
    pol = "actor.actor.policy."
    rd = rewards.to(action.dtype)
    one_minus_d = (1 - dones).to(action.dtype)
    out_grads = {}

    def emb(net, o):
        return _emb(P, net, o, goal, spec)

    cache = {}

    def emb_c(net, o, key):  # de-duplicated encoder calls
        if faithful:
            return emb(net, o)
        if (net, key) not in cache:
            cache[(net, key)] = emb(net, o)
        return cache[(net, key)]

    # ---- actor + alpha (compute_actor_and_alpha_loss :439-468)
    s_a = emb_c("actor.", obs, "obs")
    po = policy(P, pol, s_a, discrete_gripper=dg)
    mu, std = po[0], po[1]
    z = mu + noise["eps_pi"] * std
    cur_a, log_pi = torch.tanh(z), tanh_logprob(z, mu, std)
    if dg:
        nl = po[2] - po[2].logsumexp(-1, keepdim=True)
        gi = gumbel_rsample_hard_index(nl, noise["g_pi"])
        log_pi = log_pi + gripper_logprob(po[2], gi)
        cur_a = torch.cat([cur_a, gi.unsqueeze(-1).to(cur_a.dtype) * 2.0 - 1], dim=-1)
    alpha_loss = -(P["log_alpha"][0] * (log_pi + spec.target_entropy).detach()).mean()
    g = _grads(alpha_loss, P, ["log_alpha"])
    out_grads.update(g)
    opts["alpha"].step(P, g)
    alpha = P["log_alpha"][0].exp()
    logs["alpha"] = alpha.item()
    if epoch < spec.bc_epochs:
        if dg:
            lp = tanh_logprob_of_value(action[..., :-1], mu, std) + gripper_logprob(
                po[2], action[..., -1] / 2 + 0.5)
        else:
            lp = tanh_logprob_of_value(action, mu, std)
        actor_loss = (alpha * log_pi - lp).mean()
    else:
        qv = torch.min(qnet(P, "q1.critic.Q.", emb_c("q1.", obs, "obs"), cur_a),
                       qnet(P, "q2.critic.Q.", emb_c("q2.", obs, "obs"), cur_a))
        actor_loss = (alpha * log_pi - qv).mean()

    # ---- critic target (compute_critic_loss :284-314)
    with torch.no_grad():
        s_an = emb("actor.", next_obs)
        pn = policy(P, pol, s_an, discrete_gripper=dg)
        zn = pn[0] + noise["eps_next"] * pn[1]
        nxt_a, nxt_lp = torch.tanh(zn), tanh_logprob(zn, pn[0], pn[1])
        if dg:
            nln = pn[2] - pn[2].logsumexp(-1, keepdim=True)
            gn = gumbel_argmax(nln, noise["g_next"])
            nxt_lp = nxt_lp + gripper_logprob(pn[2], gn)
            nxt_a = torch.cat([nxt_a, gn.unsqueeze(-1).to(nxt_a.dtype) * 2.0 - 1], dim=-1)
        qn = torch.min(qnet(P, "target_q1.critic.Q.", emb("target_q1.", next_obs), nxt_a),
                       qnet(P, "target_q2.critic.Q.", emb("target_q2.", next_obs), nxt_a))
        if not spec.deterministic_backup:
            qn = qn - alpha * nxt_lp
        y = spec.reward_scale * rd + one_minus_d * spec.discount * qn
    s1, s2 = emb_c("q1.", obs, "obs"), emb_c("q2.", obs, "obs")
    q1_pred, q2_pred = qnet(P, "q1.critic.Q.", s1, action), qnet(P, "q2.critic.Q.", s2, action)
    bell1, bell2 = F.mse_loss(q1_pred, y), F.mse_loss(q2_pred, y)

    # ---- conservative term (compute_conservative_loss :316-406)
    A = cur_a.shape[-1]
    rand_a = noise["u_rand"] * 2.0 - 1.0
    if dg:
        rand_a = rand_a.clone()
        rand_a[..., -1] = torch.where(rand_a[..., -1] >= 0, 1.0, -1.0)
    rand_density = math.log(0.5 ** A)

    def q_both(acts_flat):  # (nB,A) -> two (B,n)
        e1 = _expand(emb("q1.", obs), n) if faithful else _expand(s1, n)
        e2 = _expand(emb("q2.", obs), n) if faithful else _expand(s2, n)
        qa = qnet(P, "q1.critic.Q.", e1, acts_flat).view(n, B).transpose(0, 1)
        qb = qnet(P, "q2.critic.Q.", e2, acts_flat).view(n, B).transpose(0, 1)
        return qa, qb

    def sample_n(po_, eps, gu):
        with torch.no_grad():
            zz = po_[0].unsqueeze(0) + eps * po_[1].unsqueeze(0)
            aa, lp_ = torch.tanh(zz), tanh_logprob(zz, po_[0], po_[1])
            if dg:
                nl_ = po_[2] - po_[2].logsumexp(-1, keepdim=True)
                gi_ = gumbel_argmax(nl_.unsqueeze(0).expand(n, -1, -1), gu)
                lp_ = lp_ + gripper_logprob(po_[2].unsqueeze(0).expand(n, -1, -1), gi_)
                aa = torch.cat([aa, gi_.unsqueeze(-1).to(aa.dtype) * 2 - 1], dim=-1)
        return aa.reshape(-1, aa.shape[-1]), lp_.squeeze(-1).transpose(0, 1)

    q1_rand, q2_rand = q_both(rand_a)
    if faithful:
        with torch.no_grad():
            po_c = policy(P, pol, emb("actor.", obs), discrete_gripper=dg)
            pn_c = policy(P, pol, emb("actor.", next_obs), discrete_gripper=dg)
    else:
        po_c, pn_c = [t.detach() for t in po], pn
    a_cur, lp_cur = sample_n(po_c, noise["eps_cur"], noise.get("g_cur"))
    a_nxt, lp_nxt = sample_n(pn_c, noise["eps_nxt"], noise.get("g_nxt"))
    q1_cur, q2_cur = q_both(a_cur)
    q1_nxt, q2_nxt = q_both(a_nxt)
    q1_data = qnet(P, "q1.critic.Q.", emb("q1.", obs), action) if faithful else q1_pred
    q2_data = qnet(P, "q2.critic.Q.", emb("q2.", obs), action) if faithful else q2_pred
    logs.update(q1_data=q1_data.mean().item(), q1_random=q1_rand.mean().item(),
                q1_policy=q1_cur.mean().item(), q2_data=q2_data.mean().item(),
                q2_random=q2_rand.mean().item(), q2_policy=q2_cur.mean().item())
    cat1 = torch.cat([q1_rand - rand_density, q1_cur - lp_cur, q1_nxt - lp_nxt], dim=1)
    cat2 = torch.cat([q2_rand - rand_density, q2_cur - lp_cur, q2_nxt - lp_nxt], dim=1)
    w, temp = spec.conservative_weight, spec.temp
    cons1 = torch.logsumexp(cat1 / temp, dim=1).mean() * w * temp - q1_data.mean() * w
    cons2 = torch.logsumexp(cat2 / temp, dim=1).mean() * w * temp - q2_data.mean() * w
    if spec.with_lagrange:
        alpha_p = torch.clamp(P["log_alpha_prime"][0].exp(), min=0.0, max=1000000.0)
        logs["alpha_prime"] = alpha_p.item()
        cons1 = alpha_p * (cons1 - spec.lagrange_thresh)
        cons2 = alpha_p * (cons2 - spec.lagrange_thresh)
        ap_loss = (-cons1 - cons2) * 0.5
        logs["alpha_prime_loss"] = ap_loss.item()
        g = _grads(ap_loss, P, ["log_alpha_prime"])
        out_grads.update(g)
    q1_loss, q2_loss = bell1 + cons1, bell2 + cons2
    logs.update(bellman_q1_loss=bell1.item(), conservative_q1_loss=cons1.item(), q1_loss=q1_loss.item(),
                bellman_q2_loss=bell2.item(), conservative_q2_loss=cons2.item(), q2_loss=q2_loss.item(),
                actor_loss=actor_loss.item(), alpha_loss=alpha_loss.item())

    # ---- all grads are taken on the pre-step graph (the reference's retain_graph
    # dance, :519-538, gives exactly these because no loss graph reads a parameter
    # that an earlier optimiser step of the same iteration mutated - SURVEY 8a note 2/6)
    an, q1n, q2n = group_names(P, "actor."), group_names(P, "q1."), group_names(P, "q2.")
    ga, g1, g2 = _grads(actor_loss, P, an), _grads(q1_loss, P, q1n), _grads(q2_loss, P, q2n, retain=False)
    if spec.with_lagrange:
        opts["alpha_prime"].step(P, {"log_alpha_prime": out_grads["log_alpha_prime"]})
    for names, gg, key in ((an, ga, "actor"), (q1n, g1, "q1"), (q2n, g2, "q2")):
        out_grads.update({k: v.clone() for k, v in gg.items()})
        clip_grads_(gg, names, spec.clip_grad_val)
        opts[key].step(P, gg)
    with torch.no_grad():  # soft_update_from_to :229-232
        for src, dst in (("q1.", "target_q1."), ("q2.", "target_q2.")):
            for nme in group_names(P, src):
                t = P[dst + nme[len(src):]]
                t.copy_(t * (1.0 - spec.tau) + P[nme] * spec.tau)
    return out_grads


def make_opts(P, spec):
    """configure_optimizers order [alpha, actor, q1, q2, alpha', AD]
    (cql_offline_lightning.py:553-574, tacorl.py:289-300)."""
    o = {
        "alpha": Adam(["log_alpha"], spec.actor_lr),
        "actor": Adam(group_names(P, "actor."), spec.actor_lr),
        "q1": Adam(group_names(P, "q1."), spec.critic_lr),
        "q2": Adam(group_names(P, "q2."), spec.critic_lr),
    }
    if spec.with_lagrange:
        o["alpha_prime"] = Adam(["log_alpha_prime"], spec.critic_lr)
    if spec.finetune_action_decoder:
        o["action_decoder"] = Adam(group_names(P, "action_decoder."), spec.action_decoder_lr)
    return o


def require_grad_(P, frozen_prefixes=()):
    for n, t in P.items():
        t.requires_grad_(not any(n.startswith(f) for f in frozen_prefixes))
    return P


# ------------------------------------------------------------------------------- A10
def pr_latent_plan(P, states, spec, eps_pr):
    """TACORL.get_pr_latent_plan (tacorl.py:235-252): frozen encoder over B*T frames ->
    posterior -> sampled plan tanh(mu + eps*std).  Returns plan, per-cam embeddings."""
    with torch.no_grad():
        emb = {}
        for c, v in states.items():
            B, T = v.shape[:2]
            emb[c] = encoder_fwd(P, f"perceptual_encoder.networks.{c}.", v.reshape(B * T, *v.shape[2:])).view(B, T, -1)
        pr_in = torch.cat([emb[c] for c in spec.pr_cams], dim=-1)
        mu, std = plan_recognition(P, "plan_recognition.", pr_in)
        return torch.tanh(mu + eps_pr * std), emb


def tacorl_step(P, opts, spec, batch, noise, epoch, faithful=False):
    """TACORL.training_step (tacorl.py:254-273).  Returns (logs, latent_plan, grads)."""
    logs = {}
    plan, emb = pr_latent_plan(P, batch["states"], spec, noise["eps_pr"])
    grads = {}
    ad_in = torch.cat([emb[c] for c in spec.ac_cams], dim=-1)[:, :-1]
    lp, ls, mm, gr = action_decoder_fwd(P, "action_decoder.", plan, ad_in)
    ad_loss = logistic_mixture_loss(lp, ls, mm, gr, batch["actions"][:, :-1])
    logs["action_loss"] = ad_loss.item()
    if spec.finetune_action_decoder:
        names = group_names(P, "action_decoder.")
        g = _grads(ad_loss, P, names, retain=False)
        grads.update(g)
        opts["action_decoder"].step(P, g)
    # get_rl_batch (tacorl.py:142-179): s=states[:,0], s'=states[:,-1], a=plan, r=d=[disp==1]
    obs = {c: v[:, 0] for c, v in batch["states"].items()}
    nxt = {c: v[:, -1] for c, v in batch["states"].items()}
    r = (batch["disp"] == 1).long().unsqueeze(-1)
    grads.update(ac_update(P, opts, spec, obs, batch["goal"], nxt, plan, r, r.clone(), noise, epoch,
                           logs, faithful=faithful))
    return logs, plan, grads


def cql_step(P, opts, spec, batch, noise, epoch, faithful=False):
    """CQL_Offline.training_step (cql_offline_lightning.py:118-147,544-551)."""
    logs = {}
    o, nx = batch["observations"], batch["next_observations"]
    r = batch["rewards"].float().unsqueeze(-1)
    d = batch["terminals"].int().unsqueeze(-1)
    grads = ac_update(P, opts, spec, o["observation"], o["goal"], nx["observation"],
                      batch["actions"].float(), r, d, noise, epoch, logs, faithful=faithful)
    return logs, grads


# ------------------------------------------------------------------------------- A13
def balanced_kl(mu_q, std_q, mu_p, std_p, kl_alpha=0.8):
    """PlayLMP.compute_kl_loss (play_lmp_for_rl.py:259-301) on the underlying Normals:
    alpha*KL(sg(post)||prior) + (1-alpha)*KL(post||sg(prior)), mean over batch."""

    def kl(m1, s1, m2, s2):  # torch.distributions.kl._kl_normal_normal, summed over dims
        var_ratio = (s1 / s2) ** 2
        t1 = ((m1 - m2) / s2) ** 2
        return (0.5 * (var_ratio + t1 - 1 - var_ratio.log())).sum(-1)

    return (kl_alpha * kl(mu_q.detach(), std_q.detach(), mu_p, std_p).mean()
            + (1 - kl_alpha) * kl(mu_q, std_q, mu_p.detach(), std_p.detach()).mean())


def playlmp_step(P, opt, batch, noise, cams, kl_beta=1e-3, kl_alpha=0.8, step=True, dropout_p=0.0, extra=None):
    """PlayLMP.training_step + Adam (play_lmp_for_rl.py:200-257,307-317,362-368); with dropout_p > 0 the
    plan recognition runs in train mode with the keep masks noise['dropout'] (see plan_recognition).
    noise: 'eps_plan' (B,A) rsample; 'rand' list of 4 U(0,1) (B,T-1,6[,10]) for the two
    logging-only _sample calls; 'u_plan' (B,A), 'u_goal' (B,G) U(0,1)."""
    logs = {}
    st = batch["states"]
    emb = {}
    for c in cams:
        B, T = st[c].shape[:2]
        emb[c] = encoder_fwd(P, f"perceptual_encoder.networks.{c}.", st[c].reshape(B * T, *st[c].shape[2:])).view(B, T, -1)
    cat = torch.cat([emb[c] for c in cams], dim=-1)
    pp_goal = goal_encoder(P, "goal_encoder.", cat[:, -1])
    mu_p, std_p = policy(P, "plan_proposal.policy.", torch.cat([cat[:, 0], pp_goal], dim=-1))
    mu_q, std_q = plan_recognition(P, "plan_recognition.", cat,
                                   dropout=(dropout_p, noise["dropout"]) if dropout_p > 0 else None)
    kl = balanced_kl(mu_q, std_q, mu_p, std_p, kl_alpha)
    logs["kl_loss"], logs["kl_loss_scaled"] = kl.item(), (kl * kl_beta).item()
    plan = torch.tanh(mu_q + noise["eps_plan"] * std_q)
    acts = batch["actions"][:, :-1]
    lp, ls, mm, gr = action_decoder_fwd(P, "action_decoder.", plan, cat[:, :-1])
    action_loss = logistic_mixture_loss(lp, ls, mm, gr, acts)
    logs["action_loss"] = action_loss.item()
    with torch.no_grad():
        pred = logistic_sample(lp, ls, mm, gr, noise["rand"][0], noise["rand"][1])
        logs["gripper_accuracy"] = (torch.where(pred[..., -1] > 0, 1.0, -1.0) == acts[..., -1]).float().mean().item()
        rplan = noise["u_plan"] * 2.0 - 1.0
        lp2, ls2, mm2, gr2 = action_decoder_fwd(P, "action_decoder.", rplan, cat[:, :-1])
        logs["random_plan_action_loss"] = logistic_mixture_loss(lp2, ls2, mm2, gr2, acts).item()
        pred2 = logistic_sample(lp2, ls2, mm2, gr2, noise["rand"][2], noise["rand"][3])
        logs["random_plan_gripper_accuracy"] = (torch.where(pred2[..., -1] > 0, 1.0, -1.0) == acts[..., -1]).float().mean().item()
    total = kl * kl_beta + action_loss
    logs["total_loss"] = total.item()
    names = [n for n, t in P.items() if t.requires_grad]
    if extra is not None:  # (tests: the gradient that enters the encoders' backward, (B, T, 32 * cams))
        extra["d_emb"] = torch.autograd.grad(total, cat, retain_graph=True)[0].detach()
    g = _grads(total, P, names, retain=False)
    if step:
        opt.step(P, g)
    return logs, g
