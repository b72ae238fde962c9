"""Parameter initialisers matching the reference modules' torch defaults (so a freshly built
module starts from the same distribution as the reference's): nn.Conv2d / nn.Linear
kaiming-uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)) for weight and bias, the +-1e-3 heads of
MLPPolicy / MLPQNetwork (reference actor.py:245-250, critic.py:86-87), temperature 1
(visual_encoders/utils.py:32-36), LayerNorm (1, 0), nn.Embedding N(0,1), MultiheadAttention
xavier-uniform in_proj with zero biases, nn.RNN U(+-1/sqrt(hidden))."""
import math

import torch


@torch.no_grad()
def init_views_(views, gen=None, rnn_hidden=None):
    fan = {}
    for name, t in views.items():
        if name.endswith("weight") and t.dim() >= 2:
            fan[name[: -len("weight")]] = t[0].numel()
    for name, t in views.items():
        leaf = name.rsplit(".", 1)[-1]
        pre = name[: -len(leaf)]

        def U(b, t=t):
            t.copy_(((torch.rand(t.shape, generator=gen) * 2 - 1) * b).to(t.device))

        if leaf == "temperature":
            t.fill_(1.0)
        elif ".rnn." in name or name.startswith("rnn."):
            U(1.0 / math.sqrt(rnn_hidden))
        elif any(k in pre for k in ("fc_mean.", "fc_log_std.", "gripper_action.", "critic.Q.out.", "Q.out.")):
            U(1e-3)
        elif "norm" in pre and leaf == "weight":
            t.fill_(1.0)
        elif "norm" in pre and leaf == "bias":
            t.zero_()
        elif name.endswith("position_embeddings.weight"):
            t.copy_(torch.randn(t.shape, generator=gen).to(t.device))
        elif leaf == "in_proj_weight":
            U(math.sqrt(6.0 / (t.shape[0] + t.shape[1])))
        elif leaf == "in_proj_bias" or name.endswith("out_proj.bias"):
            t.zero_()
        elif leaf == "weight":
            U(1.0 / math.sqrt(t[0].numel()))
        elif leaf == "bias":
            U(1.0 / math.sqrt(fan.get(pre, t.numel())))
        else:
            t.zero_()
