"""Flat parameter blocks and their reference-named views.

Every network keeps its parameters in ONE contiguous fp32 buffer laid out the way the HIP
kernels read it (conv weights [CO][KH][KW][CI], every tensor 4-float aligned); gradients,
Adam moments and Polyak targets are buffers of the same layout, so clip/Adam/soft-update and
the RCCL all-reduce are single flat operations.  The reference's state-dict names and logical
shapes (OIHW conv weights, SURVEY 8a note 9) are exposed as *views* into the block.
"""
import torch

from . import ops

ENC_NAMES = ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias", "model.4.weight", "model.4.bias",
             "model.6.temperature", "fc_layers.0.weight", "fc_layers.0.bias", "fc_layers.3.weight",
             "fc_layers.3.bias"]
ENC_SHAPES = [(32, 8, 8, 3), (32,), (64, 4, 4, 32), (64,), (64, 3, 3, 64), (64,), (1,), (256, 128), (256,), (32, 256),
              (32,)]  # storage shapes (conv: OHWI)


def encoder_views(flat, base=0):
    """{reference name: view} over flat[base : base+size]; conv weights appear as OIHW."""
    offs, _ = ops.encoder_param_layout()
    v = {}
    for name, shp, o in zip(ENC_NAMES, ENC_SHAPES, offs):
        n = 1
        for s in shp:
            n *= s
        t = flat[base + o: base + o + n].view(*shp)
        v[name] = t.permute(0, 3, 1, 2) if len(shp) == 4 else t
    return v


def encoder_size():
    return ops.encoder_param_layout()[1]


def mlp_views(flat, base, dims, names):
    """names: per layer (weight_name, bias_name)."""
    wo, bo, _ = ops.mlp_param_layout(dims)
    v = {}
    for l, (wn, bn) in enumerate(names):
        o, i = dims[l + 1], dims[l]
        v[wn] = flat[base + wo[l]: base + wo[l] + o * i].view(o, i)
        v[bn] = flat[base + bo[l]: base + bo[l] + o]
    return v


def mlp_size(dims):
    return ops.mlp_param_layout(dims)[2]


def head_views(flat, base, dims, hidden, parts):
    """Policy MLP whose last layer is the concatenation of several reference heads
    (fc_mean | fc_log_std | gripper_action): parts = [(name, rows)] in storage order."""
    wo, bo, _ = ops.mlp_param_layout(dims)
    v = {}
    l = len(dims) - 2
    r0 = 0
    for name, rows in parts:
        v[name + ".weight"] = flat[base + wo[l] + r0 * hidden: base + wo[l] + (r0 + rows) * hidden].view(rows, hidden)
        v[name + ".bias"] = flat[base + bo[l] + r0: base + bo[l] + r0 + rows]
        r0 += rows
    return v


@torch.no_grad()
def load_named(views, P, prefix=""):
    for n, t in views.items():
        t.copy_(P[prefix + n].to(t.device))


class TensorBlock:
    """Generic flat block: a list of (reference name, shape) tensors, each 4-float aligned."""

    def __init__(self, spec, device, trainable=True):
        self.spec = list(spec)
        self.off = {}
        off = 0
        for name, shape in self.spec:
            n = 1
            for s in shape:
                n *= s
            self.off[name] = (off, tuple(shape), n)
            off = (off + n + 3) // 4 * 4
        self.size = off
        self.param = torch.zeros(off, device=device)
        self.views = {k: self.param[o: o + n].view(*shp) for k, (o, shp, n) in self.off.items()}
        self.trainable = trainable
        if trainable:
            self.grad, self.m, self.v = (torch.zeros(off, device=device) for _ in range(3))
            self.step = torch.zeros(1, dtype=torch.int32, device=device)
            self.grad_views = {k: self.grad[o: o + n].view(*shp) for k, (o, shp, n) in self.off.items()}

    def rebind_grad(self, flat):
        """Move the gradient block into caller-provided storage (a slice of a gradient arena: one all-reduce for
        several blocks)."""
        assert self.trainable and flat.numel() == self.size and flat.is_contiguous()
        flat.copy_(self.grad)
        self.grad = flat
        self.grad_views = {k: flat[o: o + n].view(*shp) for k, (o, shp, n) in self.off.items()}

    def views_of(self, flat):
        """Reference-named views into another flat buffer of this layout (Adam moments, ...)."""
        return {k: flat[o: o + n].view(*shp) for k, (o, shp, n) in self.off.items()}

    def p(self, name, flat=None):
        """Raw device address of tensor `name` inside `flat` (default: the parameter block)."""
        return (self.param if flat is None else flat).data_ptr() + 4 * self.off[name][0]

    def g(self, name):
        return self.p(name, self.grad)
