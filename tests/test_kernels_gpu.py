"""Kernel-level parity: every C-ABI op against the CPU oracle / plain torch fp32 on the
same seeded inputs.  f32 MFMA mode must meet the north-star tolerance (1e-4 rel); the bf16
MFMA mode is checked at bf16 operand precision (tolerance stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4
TOL_BF16 = 3e-2  # bf16 operands (8-bit mantissa), fp32 accumulate
# bf16 kernels vs the oracle evaluated with the same operand rounding (oracle.operand_rounding): what is left is
# fp32 summation order, transcendental-function differences and the few places where a kernel rounds one step
# earlier or later than "at the operand" (e.g. a bias gradient summed from unrounded dZ)
FWD_BF16_ROUNDED = 2e-3
GRAD_BF16_ROUNDED = 1e-2


def _dev():
    from tacorl_amd import _lib

    _lib.call("tacorl_hip_init", 0)
    return torch.device("cuda:0")


def relerr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("compute,tol", [(0, TOL_F32), (1, TOL_BF16)])
@pytest.mark.parametrize("M,K,N,act", [(37, 64, 256, 2), (300, 71, 256, 2), (5, 256, 1, 0), (130, 128, 32, 1),
                                        (64, 48, 2048, 1)])
def test_linear_fwd(compute, tol, M, K, N, act):
    from tacorl_amd import ops

    dev = _dev()
    ld = (K + 3) // 4 * 4
    x = torch.zeros(M, ld)
    x[:, :K] = rnd(M, K, seed=1)
    w, b = rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(N, seed=3, scale=0.1)
    z = F.linear(x[:, :K], w, b)
    ref = [z, F.relu(z), F.silu(z)][act]
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)  # keep alive: the ABI takes raw pointers
    y = torch.empty(M, N, device=dev)
    zz = torch.empty(M, N, device=dev)
    ops.call("tacorl_linear_fwd", 1, ops.ptr_array([xd]), ld, ops.ptr_array([wd]), ops.ptr_array([bd]),
             ops.ptr_array([y]), ops.ptr_array([zz]), ops.int_array([M]), K, N, act, compute, ops.stream())
    torch.cuda.synchronize()
    assert relerr(y, ref) < tol
    assert relerr(zz, z) < tol


@pytest.mark.parametrize("compute,tol", [(0, TOL_F32), (1, TOL_BF16)])
@pytest.mark.parametrize("geom", [(84, 84, 3, 8, 4, 32), (20, 20, 32, 4, 2, 64), (9, 9, 64, 3, 1, 64),
                                  (150, 200, 3, 8, 4, 32)])
def test_conv_fwd(compute, tol, geom):
    from tacorl_amd import ops

    dev = _dev()
    H, W, Ci, Kk, S, CO = geom
    n = [3, 2]
    xs = [rnd(k, Ci, H, W, seed=10 + i) for i, k in enumerate(n)]
    w, b = rnd(CO, Ci, Kk, Kk, seed=4, scale=1 / math.sqrt(Ci * Kk * Kk)), rnd(CO, seed=5, scale=0.1)
    refs = [F.relu(F.conv2d(x, w, b, stride=S)).permute(0, 2, 3, 1) for x in xs]
    xd = [x.permute(0, 2, 3, 1).contiguous().to(dev) for x in xs]
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev)
    ys = ops.conv2d_relu_fwd(xd, [wd, wd], [b.to(dev)] * 2, S, compute)
    torch.cuda.synchronize()
    for y, r in zip(ys, refs):
        assert relerr(y, r) < tol
    if Ci == 3:  # bf16 image storage (the bench's input format)
        xb = [x.to(torch.bfloat16) for x in xd]
        refs_b = [F.relu(F.conv2d(x.to(torch.bfloat16).float(), w, b, stride=S)).permute(0, 2, 3, 1) for x in xs]
        ys = ops.conv2d_relu_fwd(xb, [wd, wd], [b.to(dev)] * 2, S, compute)
        torch.cuda.synchronize()
        for y, r in zip(ys, refs_b):
            assert relerr(y, r) < tol


def _enc_params(seed):
    from tacorl_amd import synth

    names = ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias", "model.4.weight", "model.4.bias",
             "model.6.temperature", "fc_layers.0.weight", "fc_layers.0.bias", "fc_layers.3.weight", "fc_layers.3.bias"]
    shapes = [(32, 3, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,), (1,), (256, 128), (256,), (32, 256),
              (32,)]
    return {n: synth.param_values(n, s, seed) for n, s in zip(names, shapes)}


@pytest.mark.parametrize("compute,tol", [(0, TOL_F32), (1, TOL_BF16)])
@pytest.mark.parametrize("H,W", [(84, 84), (128, 128), (150, 200)])
def test_encoder_fwd_bwd(compute, tol, H, W):
    from oracle import tacorl_oracle as O
    from tacorl_amd import blocks, ops

    dev = _dev()
    n = [5, 3]
    flats, grads, imgs_d, outs, acts, douts, refs = [], [], [], [], [], [], []
    for i, k in enumerate(n):
        P = {kk: v.clone().requires_grad_(True) for kk, v in _enc_params(20 + i).items()}
        img = rnd(k, 3, H, W, seed=30 + i)
        # bf16 mode is held to the oracle evaluated with the MFMA's operand rounding (bf16 operands, fp32 accumulate)
        with O.operand_rounding(torch.bfloat16 if compute == 1 else None):
            out = O.encoder_fwd(P, "", img)
            dout = rnd(k, 32, seed=40 + i)
            (out * dout).sum().backward()
        flat = torch.zeros(blocks.encoder_size(), device=dev)
        blocks.load_named(blocks.encoder_views(flat), {kk: v.detach() for kk, v in P.items()})
        flats.append(flat)
        grads.append(torch.full_like(flat, float("nan")))
        imgs_d.append(img.permute(0, 2, 3, 1).contiguous().to(dev))
        outs.append(torch.empty(k, 32, device=dev))
        acts.append(torch.empty(ops.encoder_act_layout(k, H, W)[1], device=dev))
        douts.append(dout.to(dev))
        refs.append((out.detach(), {kk: v.grad for kk, v in P.items()}))
    ops.encoder_fwd(imgs_d, flats, outs, acts, H, W, compute)
    ops.encoder_bwd(imgs_d, flats, acts, douts, grads, H, W, compute)
    torch.cuda.synchronize()
    ftol, gtol = (tol, tol) if compute == 0 else (FWD_BF16_ROUNDED, GRAD_BF16_ROUNDED)
    for i in range(len(n)):
        assert relerr(outs[i], refs[i][0]) < ftol, ("forward", relerr(outs[i], refs[i][0]))
        gv = blocks.encoder_views(grads[i])
        for name, g in refs[i][1].items():
            e = relerr(gv[name], g)
            assert e < gtol, (name, e)


@pytest.mark.parametrize("compute,tol", [(0, TOL_F32), (1, TOL_BF16)])
@pytest.mark.parametrize("dims,acts", [([64, 256, 256, 256, 32], [2, 2, 2, 0]), ([80, 256, 256, 256, 1], [2, 2, 2, 0]),
                                       ([71, 256, 256, 256, 1], [2, 2, 2, 0]), ([32, 256, 256, 32], [1, 1, 0])])
def test_mlp_fwd_bwd(compute, tol, dims, acts):
    from tacorl_amd import blocks, ops

    dev = _dev()
    Ms = [70, 9]
    L = len(dims) - 1
    ld = (dims[0] + 3) // 4 * 4
    names = [(f"l{l}.w", f"l{l}.b") for l in range(L)]
    xs, flats, grads, actb, douts, dxs, refs = [], [], [], [], [], [], []
    for i, M in enumerate(Ms):
        Ws = [rnd(dims[l + 1], dims[l], seed=50 + i + l, scale=1 / math.sqrt(dims[l])).requires_grad_(True) for l in range(L)]
        bs = [rnd(dims[l + 1], seed=60 + i + l, scale=0.1).requires_grad_(True) for l in range(L)]
        x = rnd(M, dims[0], seed=70 + i).requires_grad_(True)
        h = x
        for l in range(L):
            h = F.linear(h, Ws[l], bs[l])
            h = [h, F.relu(h), F.silu(h)][acts[l]]
        dout = rnd(M, dims[-1], seed=80 + i)
        (h * dout).sum().backward()
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, names)
        for l in range(L):
            v[f"l{l}.w"].copy_(Ws[l].detach())
            v[f"l{l}.b"].copy_(bs[l].detach())
        xp = torch.zeros(M, ld, device=dev)
        xp[:, :dims[0]] = x.detach()
        xs.append(xp); flats.append(flat); grads.append(torch.full_like(flat, float("nan")))
        actb.append(torch.empty(ops.mlp_act_layout(M, dims, acts)[2], device=dev))
        douts.append(dout.to(dev)); dxs.append(torch.zeros(M, ld, device=dev))
        refs.append((h.detach(), x.grad, [w.grad for w in Ws], [b.grad for b in bs]))
    ops.mlp_fwd(xs, ld, flats, actb, Ms, dims, acts, compute)
    ops.mlp_bwd(xs, ld, flats, actb, douts, dims[-1], grads, dxs, ld, Ms, dims, acts, compute)
    torch.cuda.synchronize()
    for i, M in enumerate(Ms):
        yo = ops.mlp_act_layout(M, dims, acts)[1][-1]
        y = actb[i][yo: yo + M * dims[-1]].view(M, dims[-1])
        assert relerr(y, refs[i][0]) < tol, "forward"
        assert relerr(dxs[i][:, :dims[0]], refs[i][1]) < tol * 2, "dx"
        gv = blocks.mlp_views(grads[i], 0, dims, names)
        for l in range(L):
            assert relerr(gv[f"l{l}.w"], refs[i][2][l]) < tol * 2, ("dW", l)
            assert relerr(gv[f"l{l}.b"], refs[i][3][l]) < tol * 2, ("db", l)


def test_mfma_layout_identity():
    """A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md section 3)."""
    from tacorl_amd import ops

    dev = _dev()
    for compute in (0, 1):
        K = N = 64
        x = torch.eye(K, device=dev)
        w = (torch.arange(N * K, dtype=torch.float32).view(N, K) % 17 - 8).to(dev)  # exact in bf16
        y = ops.linear_fwd([x], [w], [torch.zeros(N, device=dev)], 0, compute)[0]
        torch.cuda.synchronize()
        assert torch.equal(y, w.t().contiguous())


@pytest.mark.parametrize("dg", [False, True])
def test_tanh_normal_sample(dg):
    from oracle import tacorl_oracle as O
    from tacorl_amd import ops

    dev = _dev()
    n, M, Ac = 4, 33, 6 if dg else 16
    HD = 2 * Ac + (2 if dg else 0)
    head = rnd(M, HD, seed=1, scale=3.0)
    head[0, 0], head[1, Ac] = 20.0, -9.0  # exercise the clamps
    eps = torch.randn(n, M, Ac, generator=torch.Generator().manual_seed(2))
    u = torch.rand(n, M, 2, generator=torch.Generator().manual_seed(3))
    mu = head[:, :Ac].clamp(-9, 9)
    sd = head[:, Ac:2 * Ac].clamp(-5, 2).exp()
    z = mu + eps * sd
    a_ref, lp_ref = torch.tanh(z), O.tanh_logprob(z, mu, sd).squeeze(-1)
    if dg:
        lg = head[:, 2 * Ac:]
        idx = O.gumbel_argmax((lg - lg.logsumexp(-1, keepdim=True)).unsqueeze(0).expand(n, -1, -1), u)
        lp_ref = lp_ref + O.gripper_logprob(lg.unsqueeze(0).expand(n, -1, -1), idx).squeeze(-1)
        a_ref = torch.cat([a_ref, idx.unsqueeze(-1).float() * 2 - 1], -1)
    A = Ac + (1 if dg else 0)
    ld = 24
    act = torch.zeros(n * M, ld, device=dev)
    logp = torch.empty(n * M, device=dev)
    hd_, ed_, ud_ = head.to(dev), eps.to(dev), u.to(dev)
    ops.tanh_normal_sample(hd_, HD, ed_, ud_ if dg else None, False, act, 4, ld, logp, None, n, M, Ac)
    torch.cuda.synchronize()
    assert relerr(act[:, 4:4 + A].view(n, M, A), a_ref) < 1e-5
    assert relerr(logp.view(n, M), lp_ref) < 1e-5


def test_adam_clip_polyak():
    from oracle import tacorl_oracle as O
    from tacorl_amd import ops

    dev = _dev()
    N = 100003
    P = {"w": rnd(N, seed=1)}
    tgt = rnd(N, seed=5)
    opt = O.Adam(["w"], 3e-4)
    p_d, m_d, v_d = P["w"].clone().to(dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    t_d, step = tgt.clone().to(dev), torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(3):
        g = rnd(N, seed=10 + it, scale=0.01 * (it + 1))
        gd = {"w": g.clone()}
        O.clip_grads_(gd, ["w"], 1.0)
        opt.step(P, gd)
        tgt = tgt * (1 - 0.005) + P["w"] * 0.005
        g_d = g.to(dev)
        ops.adam_step(p_d, g_d, m_d, v_d, 3e-4, 1.0, step, t_d, 0.005)
    torch.cuda.synchronize()
    assert int(step.item()) == 3
    assert (p_d.cpu() - P["w"]).abs().max().item() < 2e-7
    assert (t_d.cpu() - tgt).abs().max().item() < 2e-7


@pytest.mark.parametrize("det_backup,lagrange", [(True, True), (False, True), (False, False)])
@pytest.mark.parametrize("n", [4, 32])
def test_cql_loss(det_backup, lagrange, n):
    from tacorl_amd import _lib, ops

    dev = _dev()
    B, A = 37, 16
    R = (3 * n + 1) * B
    g = torch.Generator().manual_seed(7)
    q = [torch.randn(R, generator=g).requires_grad_(True) for _ in range(2)]
    tq = [torch.randn(B, generator=g) for _ in range(2)]
    lpc, lpn, nlp = torch.randn(n * B, generator=g) * 3, torch.randn(n * B, generator=g) * 3, torch.randn(B, generator=g)
    rew = (torch.rand(B, generator=g) < 0.3).float()
    la, lap = torch.tensor([0.2]), torch.tensor([-0.3], requires_grad=True)
    disc, rs, temp, w, gap = 0.95, 10.0, 1.0, 1.0, 5.0
    alpha = la[0].exp()
    qn = torch.min(tq[0], tq[1]) - (0 if det_backup else alpha * nlp)
    y = rs * rew + (1 - rew) * disc * qn
    losses, logs_ref = [], {}
    for i in range(2):
        qd = q[i][:B]
        qr, qc, qx = [q[i][B + gI * n * B: B + (gI + 1) * n * B].view(n, B).t() for gI in range(3)]
        cat = torch.cat([qr - math.log(0.5 ** A), qc - lpc.view(n, B).t(), qx - lpn.view(n, B).t()], 1)
        cons = torch.logsumexp(cat / temp, 1).mean() * w * temp - qd.mean() * w
        if lagrange:
            cons = lap[0].exp().clamp(0, 1e6) * (cons - gap)
        bell = F.mse_loss(qd, y)
        losses.append(bell + cons)
        logs_ref[f"bellman_q{i + 1}_loss"], logs_ref[f"conservative_q{i + 1}_loss"] = bell.item(), cons.item()
        logs_ref[f"q{i + 1}_random"], logs_ref[f"q{i + 1}_policy"] = qr.mean().item(), qc.mean().item()
    g1, = torch.autograd.grad(losses[0], q[0], retain_graph=True)
    g2, = torch.autograd.grad(losses[1], q[1], retain_graph=True)
    if lagrange:
        ap_loss = (-(losses[0] - F.mse_loss(q[0][:B], y)) - (losses[1] - F.mse_loss(q[1][:B], y))) * 0.5
        glap, = torch.autograd.grad(ap_loss, lap)
    dq = [torch.empty(R, device=dev) for _ in range(2)]
    logs = torch.zeros(32, device=dev)
    g_lap = torch.zeros(1, device=dev)
    ws = torch.empty(_lib.lib().tacorl_cql_ws_bytes(B), dtype=torch.uint8, device=dev)
    # keep every device tensor alive: the ABI takes raw pointers
    T = [t.detach().to(dev) for t in (q[0], q[1], tq[0], tq[1], lpc, lpn, nlp, rew, la, lap)]
    ops.call("tacorl_cql_loss", ops.ptr(T[0]), ops.ptr(T[1]), ops.ptr(dq[0]), ops.ptr(dq[1]), ops.ptr(T[2]),
             ops.ptr(T[3]), ops.ptr(T[4]), ops.ptr(T[5]), ops.ptr(T[6]), ops.ptr(T[7]), ops.ptr(T[7]),
             ops.ptr(T[8]), ops.ptr(T[9]) if lagrange else None, B, n, A, disc, rs, temp, w, gap, int(det_backup),
             1.0, ops.ptr(g_lap), ops.ptr(logs), ops.ptr(ws), ws.numel(), ops.stream())
    torch.cuda.synchronize()
    assert relerr(dq[0], g1) < 1e-5 and relerr(dq[1], g2) < 1e-5
    lg = dict(zip(_lib.LOG_SLOTS, logs.cpu().tolist()))
    for k, v in logs_ref.items():
        assert abs(lg[k] - v) < 1e-4 * max(1.0, abs(v)), (k, lg[k], v)
    if lagrange:
        assert abs(g_lap.item() - glap.item()) < 1e-4 * abs(glap.item())


# 128 x 128: conv1 over row bands; 150 x 200 (rgb_static of experiment=tacorl_real_world): conv1 -> conv2 through a ring of
# conv1 rows, online soft-argmax (encoder_ring.hip)
@pytest.mark.parametrize("H,W", [(84, 84), (44, 60), (128, 128), (150, 200)])
def test_encoder_fused_forward(H, W):
    """Fused bf16 inference kernel vs the CPU oracle (bf16 tolerance) and vs the generic bf16 path."""
    from oracle import tacorl_oracle as O
    from tacorl_amd import _lib, blocks, ops

    dev = _dev()
    assert _lib.lib().tacorl_encoder_fused_supported(H, W) == 1
    ring = (H, W) == (150, 200)
    n = [19, 8, 1] if not ring else [37, 17, 1]  # (ring: more than one FC chunk per workgroup under the budgets below)
    flats, imgs, outs_f, outs_g, acts, packed, refs, refs_r = [], [], [], [], [], [], [], []
    for i, k in enumerate(n):
        P = _enc_params(70 + i)
        img = rnd(k, 3, H, W, seed=80 + i)
        refs.append(O.encoder_fwd(P, "", img.to(torch.bfloat16).float()))
        with O.operand_rounding(torch.bfloat16):
            refs_r.append(O.encoder_fwd(P, "", img))
        flat = torch.zeros(blocks.encoder_size(), device=dev)
        blocks.load_named(blocks.encoder_views(flat), P)
        flats.append(flat)
        imgs.append(img.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16))
        outs_f.append(torch.full((k, 32), float("nan"), device=dev))
        outs_g.append(torch.empty(k, 32, device=dev))
        acts.append(torch.empty(ops.encoder_act_layout(k, H, W)[1], device=dev))
        packed.append(torch.empty(_lib.lib().tacorl_encoder_fused_wpk_bytes(), dtype=torch.uint8, device=dev))
    ops.call("tacorl_encoder_pack_weights", len(n), ops.ptr_array(flats), ops.ptr_array(packed), ops.stream())
    acts_f = [torch.full_like(a, float("nan")) for a in acts]
    acts_arg = lambda a: ops.ptr_array([a[0], None, a[2]])  # noqa: E731
    fmt = _lib.lib().tacorl_encoder_fused_act_format(H, W)  # 1: y1 / y2 saved as bf16 (every geometry with the LDS-resident backward), 2: fp32
    assert fmt == 1
    ops.call("tacorl_encoder_fwd_fused", len(n), ops.ptr_array(imgs), ops.ptr_array(packed), ops.ptr_array(flats),
             ops.ptr_array(outs_f), acts_arg(acts_f), ops.int_array(n), H, W, ops.stream())
    ops.encoder_fwd(imgs, flats, outs_g, acts, H, W, 1)
    torch.cuda.synchronize()
    for i in range(len(n)):
        assert torch.isfinite(outs_f[i]).all()
        assert relerr(outs_f[i], outs_g[i]) < 1e-2, ("vs generic bf16", relerr(outs_f[i], outs_g[i]))
        assert relerr(outs_f[i], refs[i]) < TOL_BF16, ("vs oracle", relerr(outs_f[i], refs[i]))
        e = relerr(outs_f[i], refs_r[i])
        assert e < FWD_BF16_ROUNDED, ("vs oracle with bf16 operand rounding", e)
    # the same launch on a workgroup budget (tacorl_encoder_fwd_fused_wg: TACORL runs the update's own problems on 192
    # workgroups beside the plan recognition): which workgroup serves which image changes nothing - bit-identical outputs
    for budget in (3, 100):
        outs_b = [torch.full((k, 32), float("nan"), device=dev) for k in n]
        acts_b = [torch.full_like(a, float("nan")) for a in acts]
        ops.call("tacorl_encoder_fwd_fused_wg", len(n), ops.ptr_array(imgs), ops.ptr_array(packed), ops.ptr_array(flats),
                 ops.ptr_array(outs_b), acts_arg(acts_b), ops.int_array(n), H, W, budget, ops.stream())
        torch.cuda.synchronize()
        for i in range(len(n)):
            assert torch.equal(outs_b[i], outs_f[i]), (budget, i)
        for i in (0, 2):
            assert torch.equal(acts_b[i].view(torch.int32), acts_f[i].view(torch.int32)), (budget, i)
    assert _lib.lib().tacorl_encoder_fwd_fused_wg(len(n), ops.ptr_array(imgs), ops.ptr_array(packed), ops.ptr_array(flats), ops.ptr_array(outs_f),
                                                  None, ops.int_array(n), H, W, 2, ops.stream()) != 0  # fewer workgroups than problems
    for i in (0, 2):  # saved activations (what tacorl_encoder_bwd reads) equal the per-layer path's
        offs, tot = ops.encoder_act_layout(n[i], H, W)
        for j, name in enumerate(["y1", "y2", "y3", "softargmax", "fc1"]):
            end = offs[j + 1] if j + 1 < 5 else tot
            a, b = acts_f[i][offs[j]:end], acts[i][offs[j]:end]
            if j < 2 and fmt == 1:  # the fused launch saves y1 / y2 as bf16 at the start of their fp32-sized slots
                a = a.view(torch.bfloat16)[: b.numel()].float()
            assert torch.isfinite(a).all(), name
            assert relerr(a, b) < 1e-2, (name, relerr(a, b))


# 128 x 128: conv1 weight gradient in 2 bands; 150 x 200: conv1 / conv2 weight gradients in 3 / 2 bands, single-buffered dgrads
@pytest.mark.parametrize("H,W", [(84, 84), (64, 64), (44, 60), (128, 128), (150, 200)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_encoder_fused_backward(H, W, accumulate):
    """Per-image LDS-resident conv backward (tacorl_encoder_bwd_fused) vs the generic bf16 path on the
    same saved activations (same bf16 operand rounding, different fp32 summation order only), and vs
    autograd of the oracle evaluated with the MFMA's operand rounding (relative error, not cosine)."""
    from oracle import tacorl_oracle as O
    from tacorl_amd import _lib, blocks, ops

    dev = _dev()
    assert _lib.lib().tacorl_encoder_fused_supported(H, W) == 1
    from tests.golden_util import gradient_floor

    n = [7, 13, 2]
    flats, imgs, outs, acts, douts, refs, floors = [], [], [], [], [], [], []
    for i, k in enumerate(n):
        P = {kk: v.clone().requires_grad_(True) for kk, v in _enc_params(120 + i).items()}
        img = rnd(k, 3, H, W, seed=130 + i).to(torch.bfloat16).float()
        dout = rnd(k, 32, seed=140 + i)
        with O.operand_rounding(torch.bfloat16):  # same algorithm, the MFMA's operand rounding
            (O.encoder_fwd(P, "", img) * dout).sum().backward()

            def regrad(Pp, img=img, dout=dout):
                gs = torch.autograd.grad((O.encoder_fwd(Pp, "", img) * dout).sum(), list(Pp.values()))
                return dict(zip(Pp, gs))

            # the soft-argmax temperature gradient is one global (dp - <p, dp>) cancellation: its reproducibility
            # under a 1-ulp perturbation of the weights bounds what a kernel can be held to (golden_util.gradient_floor)
            floors.append(gradient_floor(regrad, P, {kk: v.grad for kk, v in P.items()}))
        flat = torch.zeros(blocks.encoder_size(), device=dev)
        blocks.load_named(blocks.encoder_views(flat), {kk: v.detach() for kk, v in P.items()})
        flats.append(flat)
        imgs.append(img.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16))
        outs.append(torch.empty(k, 32, device=dev))
        acts.append(torch.empty(ops.encoder_act_layout(k, H, W)[1], device=dev))
        douts.append(dout.to(dev))
        refs.append({kk: v.grad for kk, v in P.items()})
    ops.encoder_fwd(imgs, flats, outs, acts, H, W, 1)
    base = 0.25 if accumulate else float("nan")
    g_gen = [torch.full_like(f, base) for f in flats]
    g_fus = [torch.full_like(f, base) for f in flats]
    ops.encoder_bwd(imgs, flats, acts, douts, g_gen, H, W, 1, accumulate=accumulate)
    # the fused backward reads y1 / y2 in the format the fused forward saves them: bf16 at the start of their slots
    acts_b = []
    for i, k in enumerate(n):
        offs, tot = ops.encoder_act_layout(k, H, W)
        ab = acts[i].clone()
        for j in (0, 1):
            seg = acts[i][offs[j]:offs[j + 1]]
            ab[offs[j]:offs[j + 1]].view(torch.bfloat16)[: seg.numel()] = seg.to(torch.bfloat16)
        acts_b.append(ab)
    ops.encoder_bwd(imgs, flats, acts_b, douts, g_fus, H, W, 1, accumulate=accumulate, fused=True)
    torch.cuda.synchronize()
    for i in range(len(n)):
        vg, vf = blocks.encoder_views(g_gen[i]), blocks.encoder_views(g_fus[i])
        for name, g in refs[i].items():
            assert torch.isfinite(vf[name]).all(), name
            e = relerr(vf[name] - (0.25 if accumulate else 0.0), vg[name] - (0.25 if accumulate else 0.0))
            assert e < 2e-3, ("vs generic bf16", name, e)
            e = relerr(vf[name] - (0.25 if accumulate else 0.0), g)
            assert e < max(GRAD_BF16_ROUNDED, 3 * floors[i].get(name, 0.0)), (
                "vs oracle autograd with bf16 operand rounding", name, e, floors[i].get(name))


# (150 x 200 has no single-launch conv3 stage: both settings of the switch run the same launches there - what the case adds is
# the image loop of the single-buffered dgrads and of the banded conv1 / conv2 weight gradients, and their run-to-run determinism)
@pytest.mark.parametrize("H,W,n", [(84, 84, [300, 90, 5]), (64, 64, [200, 171]), (44, 60, [260]), (150, 200, [270, 33])])
def test_encoder_fused_backward_image_loop_and_single_launch_conv3(H, W, n, monkeypatch):
    """More images than workgroups (every workgroup loops over several images through its double buffers, ragged
    counts, problems of different sizes): the conv3 stage as ONE launch (soft-argmax backward + dgrad3 + wgrad3 with wave
    roles, ebw_l3_kernel) against the same stage as four launches (TACORL_EBW_FUSE3=0) and against the generic bf16
    path - same operand rounding everywhere, fp32 summation order differs; twice, for run-to-run determinism."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    flats, imgs, outs, acts, douts = [], [], [], [], []
    for i, k in enumerate(n):
        P = _enc_params(220 + i)
        flat = torch.zeros(blocks.encoder_size(), device=dev)
        blocks.load_named(blocks.encoder_views(flat), P)
        flats.append(flat)
        imgs.append(rnd(k, H, W, 3, seed=230 + i).to(dev).to(torch.bfloat16))
        outs.append(torch.empty(k, 32, device=dev))
        acts.append(torch.empty(ops.encoder_act_layout(k, H, W)[1], device=dev))
        douts.append(rnd(k, 32, seed=240 + i).to(dev))
    ops.encoder_fwd(imgs, flats, outs, acts, H, W, 1)
    acts_b = []
    for i, k in enumerate(n):
        offs, tot = ops.encoder_act_layout(k, H, W)
        ab = acts[i].clone()
        for j in (0, 1):
            seg = acts[i][offs[j]:offs[j + 1]]
            ab[offs[j]:offs[j + 1]].view(torch.bfloat16)[: seg.numel()] = seg.to(torch.bfloat16)
        acts_b.append(ab)
    g_gen = [torch.full_like(f, float("nan")) for f in flats]
    ops.encoder_bwd(imgs, flats, acts, douts, g_gen, H, W, 1)
    res = {}
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("TACORL_EBW_FUSE3", mode)
        g = [torch.full_like(f, float("nan")) for f in flats]
        ops.encoder_bwd(imgs, flats, acts_b, douts, g, H, W, 1, fused=True)
        torch.cuda.synchronize()
        if mode in res:  # second run of the single-launch path: bit-identical (fixed reduction order, no atomics)
            for i, (a, b) in enumerate(zip(res[mode], g)):
                va, vb = blocks.encoder_views(a), blocks.encoder_views(b)
                bad = {k: int((va[k] != vb[k]).sum()) for k in va if not torch.equal(va[k], vb[k])}
                assert not bad, f"problem {i}: two runs differ in {bad}"
        res[mode] = g
    for i in range(len(n)):
        v1, v0, vg = (blocks.encoder_views(x[i]) for x in (res["1"], res["0"], g_gen))
        for name in v1:
            assert torch.isfinite(v1[name]).all(), name
            assert relerr(v1[name], v0[name]) < 2e-4, ("one launch vs four", name, relerr(v1[name], v0[name]))
            assert relerr(v1[name], vg[name]) < 2e-3, ("vs generic bf16", name, relerr(v1[name], vg[name]))


@pytest.mark.parametrize("dims,acts", [([64, 256, 256, 256, 32], [2, 2, 2, 0]), ([80, 256, 256, 256, 1], [2, 2, 2, 0]),
                                       ([32, 256, 256, 32], [1, 1, 0]), ([64, 128, 24], [2, 0])])
def test_mlp_fused_forward(dims, acts):
    """Single-launch MLP forward vs the per-layer bf16 path (same operand rounding) - outputs and every
    saved activation - and vs fp32 torch at bf16 tolerance; ragged row counts, several problems."""
    from oracle import tacorl_oracle as O
    from tacorl_amd import _lib, blocks, ops

    dev = _dev()
    Ms = [70, 9, 300]
    L = len(dims) - 1
    assert _lib.lib().tacorl_mlp_fwd_fused_supported(len(Ms), L, ops.int_array(dims), dims[0]) == 1
    xs, flats, fb, a_f, a_g, refs, refs_r = [], [], [], [], [], [], []
    for i, M in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        h = rnd(M, dims[0], seed=170 + i)
        hr = h
        xs.append(h.to(dev).contiguous())
        for l in range(L):
            W = rnd(dims[l + 1], dims[l], seed=150 + i + l, scale=1 / math.sqrt(dims[l]))
            b = rnd(dims[l + 1], seed=160 + i + l, scale=0.1)
            v[f"l{l}.w"].copy_(W); v[f"l{l}.b"].copy_(b)
            h = F.linear(h, W, b)
            h = [h, F.relu(h), F.silu(h)][acts[l]]
            with O.operand_rounding(torch.bfloat16):
                hr = O._linear(hr, W, b)
            hr = [hr, F.relu(hr), F.silu(hr)][acts[l]]
        refs.append(h)
        refs_r.append(hr)
        flats.append(flat)
        fb.append(flat.to(torch.bfloat16))
        n_act = ops.mlp_act_layout(M, dims, acts)[2]
        a_f.append(torch.zeros(n_act, device=dev))  # (4-float alignment gaps between layers stay unwritten)
        a_g.append(torch.zeros(n_act, device=dev))
    ops.mlp_fwd(xs, dims[0], flats, a_g, Ms, dims, acts, 1)
    ops.mlp_fwd(xs, dims[0], flats, a_f, Ms, dims, acts, 1, params_bf16=fb)
    torch.cuda.synchronize()
    for i, M in enumerate(Ms):
        assert torch.isfinite(a_f[i]).all()
        assert relerr(a_f[i], a_g[i]) < 2e-3, relerr(a_f[i], a_g[i])
        yo = ops.mlp_act_layout(M, dims, acts)[1][-1]
        y = a_f[i][yo: yo + M * dims[-1]].view(M, dims[-1])
        assert relerr(y, refs[i]) < TOL_BF16
        assert relerr(y, refs_r[i]) < FWD_BF16_ROUNDED, ("vs torch with bf16 operand rounding", relerr(y, refs_r[i]))


def test_mlp_fused_forward_ragged_input_width():
    """The Q head of a 7-dim-action critic: input width 64 + 7 = 71 (rows of the first weight matrix are not
    8-byte aligned in the bf16 mirror, the last 8-column chunk of the input is ragged), row pitch 72."""
    from tacorl_amd import _lib, blocks, ops

    dev = _dev()
    dims, acts, Ms, ld = [71, 256, 256, 256, 1], [2, 2, 2, 0], [70, 9, 300], 72
    L = len(dims) - 1
    assert _lib.lib().tacorl_mlp_fwd_fused_supported(len(Ms), L, ops.int_array(dims), ld) == 1
    xs, flats, fb, a_f, a_g = [], [], [], [], []
    for i, M in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=350 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=360 + i + l, scale=0.1))
        xp = torch.full((M, ld), 7.0, device=dev)  # the pad column holds garbage: it must not reach the result
        xp[:, :dims[0]] = rnd(M, dims[0], seed=370 + i).to(dev)
        xs.append(xp); flats.append(flat); fb.append(flat.to(torch.bfloat16))
        n_act = ops.mlp_act_layout(M, dims, acts)[2]
        a_f.append(torch.zeros(n_act, device=dev)); a_g.append(torch.zeros(n_act, device=dev))
    ops.mlp_fwd(xs, ld, flats, a_g, Ms, dims, acts, 1)
    ops.mlp_fwd(xs, ld, flats, a_f, Ms, dims, acts, 1, params_bf16=fb)
    torch.cuda.synchronize()
    for i in range(len(Ms)):
        assert torch.isfinite(a_f[i]).all()
        assert relerr(a_f[i], a_g[i]) < 2e-3, relerr(a_f[i], a_g[i])


@pytest.mark.parametrize("M,dims", [(16384 + 77, [71, 256, 256, 256, 1]), (20000, [80, 256, 256, 3]), (16500, [32, 256, 256, 256, 4])])
def test_mlp_persistent_forward_equals_per_block_kernels(M, dims, monkeypatch):
    """The persistent many-row forward (one resident workgroup per CU, the 256 x 256 layers' weights in registers) against
    the per-block kernels it replaces (TACORL_MLP_PERS=0): the bf16 / fp16 saves of every hidden layer bit for bit (zero rows
    up to the padded row count included), the 1 .. 4-column output layer to fp32 summation order."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    L = len(dims) - 1
    acts = [2] * (L - 1) + [0]
    ld = (dims[0] + 3) // 4 * 4
    Ms = [M, M - 4000]
    xs, flats, fb = [], [], []
    for i, m in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=700 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=710 + i + l, scale=0.1))
        xp = torch.zeros(m, ld, device=dev)
        xp[:, :dims[0]] = rnd(m, dims[0], seed=720 + i).to(dev)
        xs.append(xp); flats.append(flat); fb.append(flat.to(torch.bfloat16))
    assert ops.mlp_lean_ok(len(Ms), dims, ld, dims[-1], ld, 1)
    out = {}
    for pers in ("0", "1"):
        monkeypatch.setenv("TACORL_MLP_PERS", pers)
        bufs = [torch.full((ops.mlp_act_layout(m, dims, acts)[2],), float("nan"), device=dev) for m in Ms]
        ops.mlp_fwd(xs, ld, flats, bufs, Ms, dims, acts, 1, params_bf16=fb, lean=True)
        torch.cuda.synchronize()
        out[pers] = bufs
    for pers in ("1",):
        for i, m in enumerate(Ms):
            yo = ops.mlp_act_layout(m, dims, acts)[1][-1]
            a, b = out["0"][i], out[pers][i]
            nl = m * dims[-1]
            assert torch.isfinite(b[yo: yo + nl]).all() and relerr(b[yo: yo + nl], a[yo: yo + nl]) < 1e-5
            a, b = a.clone(), b.clone()
            a[yo: yo + nl] = 0
            b[yo: yo + nl] = 0
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (pers, i)  # every saved hidden copy (and every untouched gap)


@pytest.mark.parametrize("M,dims,want_dx", [(16384 + 77, [71, 256, 256, 256, 1], True), (20000, [80, 256, 256, 3], True),
                                            (16500, [32, 256, 256, 256, 1], False)])
def test_mlp_persistent_backward_equals_per_block_kernels(M, dims, want_dx, monkeypatch):
    """The persistent many-row input-gradient chain against the per-block kernels it replaces (TACORL_MLP_PERS_BWD=0), on
    the same forward saves: d_x and every weight / bias gradient (they consume the chain's bf16 dZ copies) - bit for bit
    with a 1-column output layer, to fp32 summation order of the rank-NL first product otherwise."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    L = len(dims) - 1
    acts = [2] * (L - 1) + [0]
    ld = (dims[0] + 3) // 4 * 4
    Ms = [M, M - 4000]
    xs, flats, fb, douts = [], [], [], []
    for i, m in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=800 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=810 + i + l, scale=0.1))
        xp = torch.zeros(m, ld, device=dev)
        xp[:, :dims[0]] = rnd(m, dims[0], seed=820 + i).to(dev)
        xs.append(xp); flats.append(flat); fb.append(flat.to(torch.bfloat16))
        douts.append(rnd(m, dims[-1], seed=830 + i).to(dev))
    bufs = [torch.zeros(ops.mlp_act_layout(m, dims, acts)[2], device=dev) for m in Ms]
    ops.mlp_fwd(xs, ld, flats, bufs, Ms, dims, acts, 1, params_bf16=fb, lean=True)
    out = {}
    for pers in ("0", "1"):
        monkeypatch.setenv("TACORL_MLP_PERS_BWD", pers)
        dx = [torch.full((m, ld), float("nan"), device=dev) for m in Ms] if want_dx else None
        gr = [torch.zeros_like(f) for f in flats]
        ops.mlp_bwd_fused_dgrad(flats, bufs, douts, dims[-1], dx, ld, Ms, dims, acts, "t_pers_bwd", lean=True)
        ops.mlp_bwd_fused_wgrad(xs, ld, bufs, douts, dims[-1], gr, Ms, dims, acts, "t_pers_bwd", lean=True)
        torch.cuda.synchronize()
        out[pers] = (dx, gr)
    exact = dims[-1] == 1
    for i, m in enumerate(Ms):
        for a, b in ((out["0"][0][i], out["1"][0][i]) if want_dx else (None, None), (out["0"][1][i], out["1"][1][i])):
            if a is None:
                continue
            a, b = a[:, :dims[0]] if a.dim() == 2 else a, b[:, :dims[0]] if b.dim() == 2 else b
            assert torch.isfinite(b).all()
            assert torch.equal(a, b) if exact else relerr(b, a) < 2e-3, (i, relerr(b, a))


@pytest.mark.parametrize("A,cams,B,lean", [(16, 1, 37, False), (32, 2, 37, False), (7, 1, 37, False), (16, 1, 210, True), (16, 1, 1700, True)])
def test_mlp_fused_forward_gathered_input(A, cams, B, lean):
    """tacorl_mlp_fwd_fused_gather: the Q head's input [enc(obs) per camera | goal_enc | action] read in place - state rows
    repeating every B rows (expand_obs on embeddings, reference utils/misc.py:132-153), actions one row each - must give
    bit for bit the activations of `copy_cols` + tacorl_mlp_fwd_fused, write the assembled rows where asked and leave the
    other problems' x_out alone.  A = 7: a ragged last segment (64 + 7 columns, row pitch 72) taken from an 8-float pitch."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    n = 3  # (B = 210, lean: 2 100 rows - the many-row kernels with 32-row workgroups, bf16 y / fp16 act' saves)
    R, E = (3 * n + 1) * B, 64 * cams
    lda = (A + 3) // 4 * 4
    dims, acts = [E + A, 256, 256, 256, 1], [2, 2, 2, 0]
    ldq, L = (E + A + 3) // 4 * 4, 4
    Ms = [R, B]
    enc = [rnd(B + 5, 32, seed=900 + j).to(dev) for j in range(cams)]      # rows [5, 5 + B) are the problem's
    gact = rnd(1000 + B * 32 * cams, seed=910).to(dev)                     # goal-encoder output at a float offset
    actions = [torch.full((R, lda), 3.0, device=dev), torch.full((B, lda), 3.0, device=dev)]
    for i, a in enumerate(actions):
        a[:, :A] = rnd(a.shape[0], A, seed=920 + i).to(dev)
    flats, fb = [], []
    for i in range(2):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=930 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=940 + i + l, scale=0.1))
        flats.append(flat); fb.append(flat.to(torch.bfloat16))
    # reference: assemble, then the plain fused forward
    xs = [torch.zeros(M, ldq, device=dev) for M in Ms]
    for x, M, a in zip(xs, Ms, actions):
        rows = torch.arange(M, device=dev) % B
        for j in range(cams):
            x[:, 32 * j: 32 * j + 32] = enc[j][5 + rows]
        x[:, 32 * cams: 64 * cams] = gact[1000:].view(B, 32 * cams)[rows]
        x[:, E: E + A] = a[:, :A]
    a_ref = [torch.zeros(ops.mlp_act_layout(M, dims, acts)[2], device=dev) for M in Ms]
    a_got = [torch.full_like(t, float("nan")) for t in a_ref]
    ops.mlp_fwd(xs, ldq, flats, a_ref, Ms, dims, acts, 1, params_bf16=fb, lean=lean)
    assert ops.mlp_fwd_gather_ok(Ms, dims, acts, ldq, 1, lean)
    segs = [[(enc[j], 5 * 32, 32, 32 * j, mod) for j in range(cams)] + [(gact, 1000, 32 * cams, 32 * cams, mod), (a, 0, lda, E, 0)]
            for mod, a in ((B, actions[0]), (0, actions[1]))]
    x_out = torch.full((R, ldq), float("nan"), device=dev)
    if lean:
        a_got = [torch.zeros_like(t) for t in a_ref]
    ops.mlp_fwd_gather(segs, [x_out, None], ldq, flats, fb, a_got, Ms, dims, acts, lean=lean)
    torch.cuda.synchronize()
    for i, M in enumerate(Ms):
        if lean:  # (hidden outputs are not saved / saved as bf16 + fp16 copies: compare the buffers as they are)
            assert torch.equal(a_got[i].view(torch.int32), a_ref[i].view(torch.int32)), i
            continue
        zo, yo, _ = ops.mlp_act_layout(M, dims, acts)
        for l in range(L):  # (alignment gaps between layers stay unwritten)
            for off in (zo[l], yo[l]):
                if off >= 0:
                    sl = slice(off, off + M * dims[l + 1])
                    assert torch.equal(a_got[i][sl], a_ref[i][sl]), (i, l)
    assert torch.equal(x_out[:, :E + A], xs[0][:, :E + A])
    assert bool((x_out[:, E + A:] == 0).all())  # pad columns of the assembled copy: zeros, never the source's garbage
    # shapes the gather refuses: a segment start that is not a multiple of 8, a pitch that is not a multiple of 4
    from tacorl_amd import _lib
    with pytest.raises(_lib.TacorlHipError):
        ops.mlp_fwd_gather([[(enc[0], 0, 32, 0, 0), (gact, 1000, 32, 20, 0)]], [None], ldq, flats[:1], fb[:1], a_got[1:], [B], dims, acts)
    with pytest.raises(_lib.TacorlHipError):
        ops.mlp_fwd_gather([[(enc[0], 0, 30, 0, 0)]], [None], ldq, flats[:1], fb[:1], a_got[1:], [B], dims, acts)


@pytest.mark.parametrize("rows", [16384 + 77, 2048 + 13])  # (2 061: d_out's seeded mean is not ~0 there - at 2 125 rows the bias
def test_mlp_fused_backward_many_rows(rows):               # sums cancel to 3 % of their l2 scale and bf16 dZ alone moves them 6 %)
    """>= 16384 rows: the 64-rows-per-workgroup input-gradient chain and the many-row weight gradients (lean mode: LDS-DMA
    over the bf16 copies the forward and the chain leave behind, mlp_wgrad_big_kernel) vs the per-layer bf16 backward on the
    same saved activations, and vs the oracle's linear layers under bf16 operand rounding.  2 048 .. 16 383 rows (round 5:
    the headline step's Q networks, 3 328 rows): the same saves and weight-gradient kernel behind 32-row workgroups."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    dims, acts, Ms = [71, 256, 256, 256, 1], [2, 2, 2, 0], [rows, 130]
    L, ld = len(dims) - 1, 72
    xs, flats, fb, act_s, act_l, douts = [], [], [], [], [], []
    for i, M in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=650 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=660 + i + l, scale=0.1))
        xp = torch.zeros(M, ld, device=dev)
        xp[:, :dims[0]] = rnd(M, dims[0], seed=670 + i).to(dev)
        xs.append(xp); flats.append(flat); fb.append(flat.to(torch.bfloat16))
        n_act = ops.mlp_act_layout(M, dims, acts)[2]
        act_s.append(torch.zeros(n_act, device=dev)); act_l.append(torch.zeros(n_act, device=dev))
        douts.append(rnd(M, dims[-1], seed=680 + i).to(dev))
    ops.mlp_fwd(xs, ld, flats, act_s, Ms, dims, acts, 1)                                # per-layer forward: everything saved
    ops.mlp_fwd(xs, ld, flats, act_l, Ms, dims, acts, 1, params_bf16=fb, lean=True)     # fused, lean
    g_ref, g_fus = [torch.zeros_like(f) for f in flats], [torch.zeros_like(f) for f in flats]
    dx_ref, dx_fus = [torch.zeros(M, ld, device=dev) for M in Ms], [torch.zeros(M, ld, device=dev) for M in Ms]
    ops.mlp_bwd(xs, ld, flats, act_s, douts, dims[-1], g_ref, dx_ref, ld, Ms, dims, acts, 1)
    ops.mlp_bwd_fused_dgrad(flats, act_l, douts, dims[-1], dx_fus, ld, Ms, dims, acts, "t_mlp_big", lean=True)
    ops.mlp_bwd_fused_wgrad(xs, ld, act_l, douts, dims[-1], g_fus, Ms, dims, acts, "t_mlp_big", lean=True)
    torch.cuda.synchronize()
    from oracle import tacorl_oracle as O
    from tests.golden_util import record_margin

    for i in range(len(Ms)):
        assert torch.isfinite(g_fus[i]).all() and torch.isfinite(dx_fus[i]).all()
        assert relerr(dx_fus[i], dx_ref[i]) < 4e-3, ("dx", i, relerr(dx_fus[i], dx_ref[i]))
        assert relerr(g_fus[i], g_ref[i]) < 4e-3, ("grads", i, relerr(g_fus[i], g_ref[i]))
        # ... and against the CPU oracle's linear layers under the MFMA's operand rounding (torch autograd): forward output,
        # input gradients and every weight / bias gradient of the many-row kernels (mlp_fused_{fwd,bwd}_big, mlp_wgrad_fused)
        v = blocks.mlp_views(flats[i], 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        Pw = {k: t.detach().cpu().clone().requires_grad_(True) for k, t in v.items()}
        x0 = xs[i][:, :dims[0]].cpu().clone().requires_grad_(True)
        h = x0
        with O.operand_rounding(torch.bfloat16):
            for l in range(L):
                h = O._linear(h, Pw[f"l{l}.w"], Pw[f"l{l}.b"])
                h = [h, F.relu(h), F.silu(h)][acts[l]]
            (h * douts[i].cpu()).sum().backward()
        yo = ops.mlp_act_layout(Ms[i], dims, acts)[1]
        out = act_l[i][yo[L - 1]: yo[L - 1] + Ms[i] * dims[-1]].reshape(Ms[i], dims[-1])
        e = relerr(out, h)
        record_margin(f"mlp many rows M={Ms[i]}: forward output", e, FWD_BF16_ROUNDED, kind="kernel vs rounded oracle")
        assert e < FWD_BF16_ROUNDED, ("forward vs rounded oracle", i, e)
        e = relerr(dx_fus[i][:, :dims[0]], x0.grad)
        record_margin(f"mlp many rows M={Ms[i]}: dx", e, GRAD_BF16_ROUNDED, kind="kernel vs rounded oracle")
        assert e < GRAD_BF16_ROUNDED, ("dx vs rounded oracle", i, e)
        gv = blocks.mlp_views(g_fus[i], 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for k, t in Pw.items():
            e = relerr(gv[k], t.grad)
            record_margin(f"mlp many rows M={Ms[i]}: d{k}", e, GRAD_BF16_ROUNDED, kind="kernel vs rounded oracle")
            assert e < GRAD_BF16_ROUNDED, (k, i, e)


@pytest.mark.parametrize("rows", [16384 + 77, 2048 + 77])
@pytest.mark.parametrize("lean", [False, True])
def test_mlp_fused_forward_many_rows(lean, rows):
    """>= 16384 rows take the 128-rows-per-workgroup instantiation (C5's Q networks: 99 k rows): against the per-layer
    bf16 path on the same inputs, ragged last block included; in lean mode the final output must agree as well."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    dims, acts, Ms, ld = [71, 256, 256, 256, 1], [2, 2, 2, 0], [rows, 300], 72  # (2 048 .. 16 383 rows: 32-row workgroups)
    L = len(dims) - 1
    xs, flats, fb, a_f, a_g = [], [], [], [], []
    for i, M in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=550 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=560 + i + l, scale=0.1))
        xp = torch.full((M, ld), 7.0, device=dev)
        xp[:, :dims[0]] = rnd(M, dims[0], seed=570 + i).to(dev)
        xs.append(xp); flats.append(flat); fb.append(flat.to(torch.bfloat16))
        n_act = ops.mlp_act_layout(M, dims, acts)[2]
        a_f.append(torch.zeros(n_act, device=dev)); a_g.append(torch.zeros(n_act, device=dev))
    ops.mlp_fwd(xs, ld, flats, a_g, Ms, dims, acts, 1)
    ops.mlp_fwd(xs, ld, flats, a_f, Ms, dims, acts, 1, params_bf16=fb, lean=lean)
    torch.cuda.synchronize()
    for i, M in enumerate(Ms):
        zo, yo, _ = ops.mlp_act_layout(M, dims, acts)
        out_f, out_g = a_f[i][yo[L - 1]: yo[L - 1] + M * dims[-1]], a_g[i][yo[L - 1]: yo[L - 1] + M * dims[-1]]
        assert torch.isfinite(out_f).all() and relerr(out_f, out_g) < 2e-3, relerr(out_f, out_g)
        for l in range(L - 1):  # saved pre-activations (and, unless lean, outputs) of the hidden layers
            zf, zg = a_f[i][zo[l]: zo[l] + M * dims[l + 1]], a_g[i][zo[l]: zo[l] + M * dims[l + 1]]
            if lean and M >= 2048:
                # many-row problems of a lean site save what the backward reads instead of fp32 z / y (mlp_big_fwd_kernel):
                # the output as bf16 [Mp][N] in the y region, act'(z) as fp16 [Mp][N] in the z region, zero rows beyond M
                N, Mp = dims[l + 1], (M + 63) // 64 * 64
                yb = a_f[i][yo[l]: yo[l] + Mp * N // 2].view(torch.bfloat16).reshape(Mp, N)
                sb = a_f[i][zo[l]: zo[l] + Mp * N // 2].view(torch.float16).reshape(Mp, N)
                zr = zg.reshape(M, N)
                sg = torch.sigmoid(zr)
                assert relerr(yb[:M].float(), (zr * sg).to(torch.bfloat16).float()) < 2e-3
                assert relerr(sb[:M].float(), sg * (1 + zr * (1 - sg))) < 2e-3
                assert not yb[M:].any()
                continue
            assert relerr(zf, zg) < 2e-3, (l, relerr(zf, zg))
            if not lean:
                yf, yg = a_f[i][yo[l]: yo[l] + M * dims[l + 1]], a_g[i][yo[l]: yo[l] + M * dims[l + 1]]
                assert relerr(yf, yg) < 2e-3, (l, relerr(yf, yg))


@pytest.mark.parametrize("dims,acts", [([71, 256, 256, 256, 1], [2, 2, 2, 0]), ([64, 256, 256, 32], [2, 2, 0]), ([32, 256, 256, 32], [1, 1, 0])])
def test_mlp_fused_lean_activations(dims, acts):
    """lean mode: the fused forward does not write the hidden layers' outputs (SiLU: the pre-activation is saved) and the
    one-launch weight gradients recompute them while staging - outputs, input gradients and weight gradients are bit for
    bit those of the saving mode, and the skipped regions of the activation buffer are really untouched."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    Ms, L = [70, 300], len(dims) - 1
    ld = (dims[0] + 3) // 4 * 4
    lean_ok = ops.mlp_lean_ok(len(Ms), dims, ld, dims[-1], ld, 1)
    assert lean_ok
    xs, flats, fb, douts = [], [], [], []
    for i, M in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=450 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=460 + i + l, scale=0.1))
        xp = torch.zeros(M, ld, device=dev)
        xp[:, :dims[0]] = rnd(M, dims[0], seed=470 + i).to(dev)
        xs.append(xp); flats.append(flat); fb.append(flat.to(torch.bfloat16)); douts.append(rnd(M, dims[-1], seed=480 + i).to(dev))
    res = []
    for lean in (False, True):
        actb = [torch.full((ops.mlp_act_layout(M, dims, acts)[2],), float("nan"), device=dev) for M in Ms]
        ops.mlp_fwd(xs, ld, flats, actb, Ms, dims, acts, 1, params_bf16=fb, lean=lean)
        g = [torch.zeros_like(f) for f in flats]
        dx = [torch.zeros(M, ld, device=dev) for M in Ms]
        ops.mlp_bwd_fused_dgrad(flats, actb, douts, dims[-1], dx, ld, Ms, dims, acts, "t_mlp_lean")
        ops.mlp_bwd_fused_wgrad(xs, ld, actb, douts, dims[-1], g, Ms, dims, acts, "t_mlp_lean", lean=lean)
        torch.cuda.synchronize()
        res.append((actb, g, dx))
    (a0, g0, d0), (a1, g1, d1) = res
    for i, M in enumerate(Ms):
        zo, yo, _ = ops.mlp_act_layout(M, dims, acts)
        assert torch.equal(a0[i][yo[L - 1]: yo[L - 1] + M * dims[-1]], a1[i][yo[L - 1]: yo[L - 1] + M * dims[-1]])  # the MLP's output
        assert torch.equal(g0[i], g1[i]) and torch.equal(d0[i], d1[i]) and torch.isfinite(g1[i]).all()
        for l in range(L - 1):
            hidden = a1[i][yo[l]: yo[l] + M * dims[l + 1]]
            if zo[l] >= 0:  # pre-activation saved -> output skipped (still the NaN fill)
                assert torch.isnan(hidden).all()
            else:           # ReLU layers save only their output: kept
                assert torch.isfinite(hidden).all()


def test_prep_and_reduce_batches_equal_separate_launches():
    """ops.prep_batch (weight-only preparation of several sites + bf16 mirrors as ONE launch) and ops.reduce_batch (the slab
    reduces of several sites' one-launch weight gradients as ONE launch) against the same calls launched one by one:
    bit-identical transposed weights / mirrors / gradients; nothing is written before the block ends; misuse is refused."""
    from tacorl_amd import _lib, blocks, ops

    dev = _dev()
    sites = [([71, 256, 256, 256, 1], [2, 2, 2, 0], [70, 300]), ([32, 256, 256, 32], [1, 1, 0], [64, 9, 33]),
             ([64, 256, 256, 256, 14], [2, 2, 2, 0], [128])]
    data = []
    for si, (dims, acts, Ms) in enumerate(sites):
        L, ld = len(dims) - 1, (dims[0] + 3) // 4 * 4
        xs, flats, actb, douts = [], [], [], []
        for i, M in enumerate(Ms):
            flat = torch.zeros(blocks.mlp_size(dims), device=dev)
            v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
            for l in range(L):
                v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=900 + 10 * si + i + l, scale=1 / math.sqrt(dims[l])))
                v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=950 + 10 * si + i + l, scale=0.1))
            xp = torch.zeros(M, ld, device=dev)
            xp[:, :dims[0]] = rnd(M, dims[0], seed=970 + 10 * si + i).to(dev)
            xs.append(xp); flats.append(flat)
            actb.append(torch.zeros(ops.mlp_act_layout(M, dims, acts)[2], device=dev))
            douts.append(rnd(M, dims[-1], seed=990 + 10 * si + i).to(dev))
        ops.mlp_fwd(xs, ld, flats, actb, Ms, dims, acts, 1)
        data.append((dims, acts, Ms, ld, xs, flats, actb, douts))

    def run(batched):
        tag = "t_batch_%d_" % int(batched)
        mirrors = [torch.full((f.numel(),), 7.0, device=dev, dtype=torch.bfloat16) for d in data for f in d[5]]
        srcs = [f for d in data for f in d[5]]
        cm_p = ops.prep_batch() if batched else contextlib.nullcontext()
        with cm_p:
            _lib.call("tacorl_to_bf16_batch", len(srcs), ops.ptr_array(srcs), ops.ptr_array(mirrors),
                      (ops.C.c_long * len(srcs))(*[f.numel() // 4 * 4 for f in srcs]), ops.stream())
            for si, (dims, acts, Ms, ld, xs, flats, actb, douts) in enumerate(data):
                ops.mlp_bwd_fused_pack(flats, Ms, dims, tag + str(si), dev)
            if batched:  # recorded, not launched: the mirrors still hold their fill value
                torch.cuda.synchronize()
                assert all(bool((m.float() == 7.0).all()) for m in mirrors)
        grads = [[torch.full_like(f, 3.0) for f in d[5]] for d in data]
        cm_r = ops.reduce_batch() if batched else contextlib.nullcontext()
        with cm_r:
            for si, (dims, acts, Ms, ld, xs, flats, actb, douts) in enumerate(data):
                ops.mlp_bwd_fused_dgrad(flats, actb, douts, dims[-1], None, ld, Ms, dims, acts, tag + str(si), prepacked=True)
                ops.mlp_bwd_fused_wgrad(xs, ld, actb, douts, dims[-1], grads[si], Ms, dims, acts, tag + str(si))
            if batched:  # the reduces are pending: no gradient element has been written yet
                torch.cuda.synchronize()
                assert all(bool((g == 3.0).all()) for gs in grads for g in gs)
        torch.cuda.synchronize()
        return mirrors, grads

    import contextlib
    saved = ops.prep_batch.enabled, ops.reduce_batch.enabled
    ops.prep_batch.enabled = ops.reduce_batch.enabled = True  # (reduce_batch is off by default: measured slower in the step)
    try:
        m0, g0 = run(False)
        m1, g1 = run(True)
    finally:
        ops.prep_batch.enabled, ops.reduce_batch.enabled = saved
    for a, b in zip(m0, m1):
        assert torch.equal(a, b)
    for ga, gb in zip(g0, g1):
        for a, b in zip(ga, gb):
            assert torch.isfinite(b).all() and torch.equal(a, b)
    lib = _lib.lib()
    assert lib.tacorl_prep_batch_end(ops.stream()) != 0 and lib.tacorl_reduce_batch_end(ops.stream()) != 0  # no open batch
    assert lib.tacorl_prep_batch_begin() == 0 and lib.tacorl_prep_batch_begin() != 0                       # no nesting
    assert lib.tacorl_prep_batch_end(ops.stream()) == 0


@pytest.mark.parametrize("dims,acts", [([64, 256, 256, 256, 32], [2, 2, 2, 0]), ([80, 256, 256, 256, 1], [2, 2, 2, 0]),
                                       ([71, 256, 256, 256, 1], [2, 2, 2, 0]), ([32, 256, 256, 32], [1, 1, 0])])
@pytest.mark.parametrize("want_dx", [True, False])
def test_mlp_fused_backward(dims, acts, want_dx):
    """Single-launch input-gradient chain + separate weight-gradient call vs the per-layer bf16 backward
    on the same saved activations (same operand rounding; fp32 summation order differs)."""
    from tacorl_amd import blocks, ops

    dev = _dev()
    Ms = [70, 9, 300]
    L = len(dims) - 1
    ld = (dims[0] + 3) // 4 * 4
    assert ops.mlp_bwd_fused_ok(len(Ms), dims, dims[-1], ld, 1)
    xs, flats, actb, douts = [], [], [], []
    for i, M in enumerate(Ms):
        flat = torch.zeros(blocks.mlp_size(dims), device=dev)
        v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        for l in range(L):
            v[f"l{l}.w"].copy_(rnd(dims[l + 1], dims[l], seed=250 + i + l, scale=1 / math.sqrt(dims[l])))
            v[f"l{l}.b"].copy_(rnd(dims[l + 1], seed=260 + i + l, scale=0.1))
        xp = torch.zeros(M, ld, device=dev)
        xp[:, :dims[0]] = rnd(M, dims[0], seed=270 + i).to(dev)
        xs.append(xp); flats.append(flat)
        actb.append(torch.zeros(ops.mlp_act_layout(M, dims, acts)[2], device=dev))
        douts.append(rnd(M, dims[-1], seed=280 + i).to(dev))
    ops.mlp_fwd(xs, ld, flats, actb, Ms, dims, acts, 1)
    g_ref = [torch.zeros_like(f) for f in flats]  # (4-float alignment gaps of the block stay unwritten)
    g_fus = [torch.zeros_like(f) for f in flats]
    dx_ref = [torch.zeros(M, ld, device=dev) for M in Ms]
    dx_fus = [torch.zeros(M, ld, device=dev) for M in Ms]
    ops.mlp_bwd(xs, ld, flats, actb, douts, dims[-1], g_ref, dx_ref if want_dx else None, ld, Ms, dims, acts, 1)
    ops.mlp_bwd_fused_dgrad(flats, actb, douts, dims[-1], dx_fus if want_dx else None, ld, Ms, dims, acts, "t_mlp_fused")
    g_fus[1] = None  # a problem without parameter gradients (the actor's pass through the Q networks)
    ops.mlp_bwd_fused_wgrad(xs, ld, actb, douts, dims[-1], g_fus, Ms, dims, acts, "t_mlp_fused")
    torch.cuda.synchronize()
    from oracle import tacorl_oracle as O

    for i in range(len(Ms)):
        if want_dx:
            assert relerr(dx_fus[i], dx_ref[i]) < 3e-3, ("dx", i, relerr(dx_fus[i], dx_ref[i]))
        if g_fus[i] is not None:
            assert torch.isfinite(g_fus[i]).all()
            assert relerr(g_fus[i], g_ref[i]) < 3e-3, ("grads", i, relerr(g_fus[i], g_ref[i]))
        # ... and not only against the repo's own per-layer path: torch autograd with the MFMA's operand rounding
        v = blocks.mlp_views(flats[i], 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
        Pw = {k: t.detach().cpu().clone().requires_grad_(True) for k, t in v.items()}
        x0 = xs[i][:, :dims[0]].cpu().clone().requires_grad_(True)
        h = x0
        with O.operand_rounding(torch.bfloat16):
            for l in range(L):
                h = O._linear(h, Pw[f"l{l}.w"], Pw[f"l{l}.b"])
                h = [h, F.relu(h), F.silu(h)][acts[l]]
            (h * douts[i].cpu()).sum().backward()
        if want_dx:
            e = relerr(dx_fus[i][:, :dims[0]], x0.grad)
            assert e < GRAD_BF16_ROUNDED, ("dx vs rounded autograd", i, e)
        if g_fus[i] is not None:
            gv = blocks.mlp_views(g_fus[i], 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
            for k, t in Pw.items():
                e = relerr(gv[k], t.grad)
                assert e < GRAD_BF16_ROUNDED, ("grad vs rounded autograd", i, k, e)


@pytest.mark.parametrize("M,K,N,act", [(256, 2048, 2048, 1), (100, 256, 64, 0), (3840, 2048, 2048, 0), (64, 128, 32, 1),
                                         (3840, 2048, 192, 0), (1030, 256, 64, 1), (2000, 384, 448, 0)])  # few N tiles over many rows: 64 x 64 tiles, plain map (the output heads)
def test_rnn_linear_bf16(M, K, N, act):
    """LDS-DMA ring GEMM of the ReLU-RNN (bf16 operands in HBM): vs fp32 torch on the same bf16-rounded
    operands (only the fp32 accumulation order differs), with bias, addend, ReLU and the bf16 output copy."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    assert _lib.lib().tacorl_rnn_linear_supported(M, K, N) == 1
    x = rnd(M, K, seed=1).to(torch.bfloat16)
    w = (rnd(N, K, seed=2) / math.sqrt(K)).to(torch.bfloat16)
    b, add = rnd(N, seed=3, scale=0.1), rnd(M, N + 4, seed=4)
    z = x.float() @ w.float().t() + b + add[:, :N]
    ref = F.relu(z) if act == 1 else z
    xd, wd, bd, addd = x.to(dev), w.to(dev), b.to(dev), add.to(dev)
    y = torch.full((M, N), float("nan"), device=dev)
    yb = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ops.call("tacorl_rnn_linear_fwd", ops.ptr(xd), ops.ptr(wd), ops.ptr(bd), ops.ptr(addd), N + 4, ops.ptr(y), ops.ptr(yb),
             M, K, N, act, ops.stream())
    torch.cuda.synchronize()
    assert relerr(y, ref) < 1e-5, relerr(y, ref)
    assert torch.equal(yb, y.to(torch.bfloat16))


@pytest.mark.parametrize("M,M2,nprob", [(32, 32, 3), (256, 256, 3), (64, 64, 2), (40, 24, 1), (3840, 3840, 1)])
def test_rnn_linear_fwd_batch_twin(M, M2, nprob):
    """Twin rows of the batched ring GEMM (PlayLMP's logging-only random-plan decoder pass riding in the real pass's
    launches): rows M.. of every problem read x2 / addend2 and write y2 / yb2 - bit-identical to running the twin rows as a
    launch of their own (same weights, same per-element accumulation order), and the first M rows unchanged."""
    from tacorl_amd import ops

    dev = _dev()
    K = N = 256 if M > 1000 else 2048
    if M > 1000:
        N = 192
    mk = lambda *s, seed, dt=torch.float32: rnd(*s, seed=seed).to(dt).to(dev)  # noqa: E731
    xs = [mk(M, K, seed=10 + p, dt=torch.bfloat16) for p in range(nprob)]
    x2 = [mk(M2, K, seed=20 + p, dt=torch.bfloat16) for p in range(nprob)]
    ws = [(rnd(N, K, seed=30 + p) / math.sqrt(K)).to(torch.bfloat16).to(dev) for p in range(nprob)]
    bs = [mk(N, seed=40 + p) for p in range(nprob)]
    ad = [mk(M, N, seed=50 + p) if p != 1 else None for p in range(nprob)]
    ad2 = [mk(M2, N, seed=60 + p) if p != 1 else None for p in range(nprob)]
    acts = [1 if p != 1 else 0 for p in range(nprob)]
    nan = lambda r, dt=torch.float32: torch.full((r, N), float("nan"), device=dev, dtype=dt)  # noqa: E731
    y, y2, yb, yb2 = ([nan(M) for _ in range(nprob)], [nan(M2) for _ in range(nprob)],
                      [nan(M, torch.bfloat16) if p != 1 else None for p in range(nprob)],
                      [nan(M2, torch.bfloat16) if p != 1 else None for p in range(nprob)])
    ops.call("tacorl_rnn_linear_fwd_batch_twin", nprob, ops.ptr_array(xs), ops.ptr_array(x2), ops.ptr_array(ws), ops.ptr_array(bs),
             ops.ptr_array(ad), ops.ptr_array(ad2), N, ops.ptr_array(y), ops.ptr_array(y2), ops.ptr_array(yb), ops.ptr_array(yb2),
             M, M2, K, N, ops.int_array(acts), ops.stream())
    for rows, xx, aa, yy, yyb in ((M, xs, ad, y, yb), (M2, x2, ad2, y2, yb2)):
        r, rb = [nan(rows) for _ in range(nprob)], [nan(rows, torch.bfloat16) if p != 1 else None for p in range(nprob)]
        ops.call("tacorl_rnn_linear_fwd_batch", nprob, ops.ptr_array(xx), ops.ptr_array(ws), ops.ptr_array(bs), ops.ptr_array(aa), N,
                 ops.ptr_array(r), ops.ptr_array(rb), rows, K, N, ops.int_array(acts), ops.stream())
        torch.cuda.synchronize()
        for p in range(nprob):
            assert torch.isfinite(yy[p]).all() and torch.equal(yy[p], r[p]), (rows, p)
            assert yyb[p] is None or torch.equal(yyb[p], rb[p]), (rows, p)
            z = xx[p].float() @ ws[p].float().t() + bs[p] + (aa[p] if aa[p] is not None else 0)
            assert relerr(yy[p], F.relu(z) if acts[p] else z) < 1e-5
    # a twin launch needs twin rows in every problem
    if nprob > 1:
        x2[0] = None
        from tacorl_amd import _lib
        rc = _lib.lib().tacorl_rnn_linear_fwd_batch_twin(
            nprob, ops.ptr_array(xs), ops.ptr_array(x2), ops.ptr_array(ws), ops.ptr_array(bs), ops.ptr_array(ad), ops.ptr_array(ad2),
            N, ops.ptr_array(y), ops.ptr_array(y2), ops.ptr_array(yb), ops.ptr_array(yb2), M, M2, K, N, ops.int_array(acts), ops.stream())
        assert rc != 0


@pytest.mark.parametrize("M,M2,nprob,Kx", [(256, 0, 3, 48), (64, 64, 2, 48), (256, 256, 3, 40), (40, 0, 1, 128)])
def test_rnn_linear_fwd_batch_ext(M, M2, nprob, Kx):
    """K extension of the batched ring GEMM (layer 0's input projection inside its recurrent step's launch): problem 0 adds
    x_ext w_ext^T + bias2 over one more K tile (operands zero padded to 128 columns), with and without twin rows; against
    fp32 torch on the same bf16-rounded operands, and the build_ad_input_bf16 rows against their fp32 originals."""
    from tacorl_amd import ops

    dev = _dev()
    K = N = 2048 if M >= 256 else 256
    mk = lambda *s, seed, dt=torch.float32: rnd(*s, seed=seed).to(dt).to(dev)  # noqa: E731
    xs = [mk(M, K, seed=10 + p, dt=torch.bfloat16) for p in range(nprob)]
    x2 = [mk(M2, K, seed=20 + p, dt=torch.bfloat16) for p in range(nprob)] if M2 else None
    ws = [(rnd(N, K, seed=30 + p) / math.sqrt(K)).to(torch.bfloat16).to(dev) for p in range(nprob)]
    bs = [mk(N, seed=40 + p) for p in range(nprob)]
    ad = [None if p == 0 else mk(M, N, seed=50 + p) for p in range(nprob)]
    ad2 = [None if p == 0 else mk(M2, N, seed=60 + p) for p in range(nprob)] if M2 else None
    acts = [1] * nprob
    pad = lambda t: torch.cat([t, torch.zeros(t.shape[0], 128 - t.shape[1], device=dev, dtype=t.dtype)], 1).contiguous()  # noqa: E731
    xe, we = pad(mk(M, Kx, seed=70, dt=torch.bfloat16)), pad((rnd(N, Kx, seed=71) / math.sqrt(Kx)).to(torch.bfloat16).to(dev))
    xe2 = pad(mk(M2, Kx, seed=72, dt=torch.bfloat16)) if M2 else None
    b2 = mk(N, seed=73)
    none = [None] * (nprob - 1)
    nan = lambda r, dt=torch.float32: torch.full((r, N), float("nan"), device=dev, dtype=dt)  # noqa: E731
    y, yb = [nan(M) for _ in range(nprob)], [nan(M, torch.bfloat16) for _ in range(nprob)]
    y2, yb2 = ([nan(M2) for _ in range(nprob)], [nan(M2, torch.bfloat16) for _ in range(nprob)]) if M2 else (None, None)
    pa = lambda v: ops.ptr_array(v) if v is not None else None  # noqa: E731
    ops.call("tacorl_rnn_linear_fwd_batch_ext", nprob, pa(xs), pa(x2), pa(ws), pa(bs), pa(ad), pa(ad2), N, pa(y), pa(y2), pa(yb), pa(yb2),
             M, M2, K, N, ops.int_array(acts), pa([xe] + none), pa([xe2] + none) if M2 else None, pa([we] + none), pa([b2] + none),
             ops.stream())
    torch.cuda.synchronize()
    for rows, xx, aa, yy, yyb, xx_e in ((M, xs, ad, y, yb, xe), (M2, x2, ad2, y2, yb2, xe2)):
        if not rows:
            continue
        for p in range(nprob):
            z = xx[p].float() @ ws[p].float().t() + bs[p] + (aa[p] if aa[p] is not None else 0)
            if p == 0:
                z = z + xx_e.float() @ we.float().t() + b2
            assert torch.isfinite(yy[p]).all() and relerr(yy[p], F.relu(z)) < 1e-5, (rows, p, relerr(yy[p], F.relu(z)))
            assert torch.equal(yyb[p], yy[p].to(torch.bfloat16))
    # the input rows: [plan | emb_t | zeros] time-major, bf16
    B, T, Tm, P, E = 6, 5, 4, 16, 32
    plan, emb = mk(B, P, seed=80), mk(B * T, E + 8, seed=81)
    out = torch.full((Tm * B, 128), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.call("tacorl_build_ad_input_bf16", ops.ptr(plan), ops.ptr(emb), E + 8, ops.ptr(out), B, T, Tm, P, E, ops.stream())
    torch.cuda.synchronize()
    exp = torch.zeros(Tm, B, 128, device=dev)
    exp[:, :, :P] = plan[None]
    exp[:, :, P:P + E] = emb.view(B, T, E + 8)[:, :Tm, :E].transpose(0, 1)
    assert torch.equal(out, exp.view(Tm * B, 128).to(torch.bfloat16))


def test_transpose_to_bf16_batch():
    """Several transposes in one launch (jobs of different shapes) against torch; refused shapes."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    shapes = [(2048, 32), (32, 2048), (64, 64), (96, 32), (256, 256)]
    srcs = [rnd(r, c, seed=40 + i).to(dev) for i, (r, c) in enumerate(shapes)]
    dsts = [torch.full((c, r), float("nan"), device=dev, dtype=torch.bfloat16) for r, c in shapes]
    ops.call("tacorl_transpose_to_bf16_batch", len(srcs), ops.ptr_array(srcs), ops.ptr_array(dsts), ops.int_array([r for r, _ in shapes]),
             ops.int_array([c for _, c in shapes]), ops.stream())
    torch.cuda.synchronize()
    for s_, d in zip(srcs, dsts):
        assert torch.equal(d, s_.t().contiguous().to(torch.bfloat16))
    rc = _lib.lib().tacorl_transpose_to_bf16_batch(1, ops.ptr_array(srcs[:1]), ops.ptr_array(dsts[:1]), ops.int_array([2048]), ops.int_array([30]), ops.stream())
    assert rc != 0


@pytest.mark.parametrize("ncam", [1, 2])
def test_playlmp_demb_assembly(ncam):
    """tacorl_ad_input_bwd(accumulate=2) defines the whole d_emb block (rows beyond the decoder's window zeroed, no fill launch)
    and tacorl_plmp_demb_finish adds the other three shares and slices the cameras - against torch, bit for bit (each element
    is a sum of at most four terms in a fixed order)."""
    from tacorl_amd import ops

    dev = _dev()
    B, T, P, Ec = 5, 6, 16, 32 * ncam
    Tm, D, D_in = T - 1, 64, Ec
    dx_seq = rnd(Tm * B, P + Ec, seed=1).to(dev)
    d_plan = torch.full((B, P), float("nan"), device=dev)
    d_emb = torch.full((B * T, Ec), float("nan"), device=dev)
    ops.call("tacorl_ad_input_bwd", ops.ptr(dx_seq), ops.ptr(d_plan), ops.ptr(d_emb), Ec, B, T, Tm, P, Ec, 2, ops.stream())
    torch.cuda.synchronize()
    x = dx_seq.view(Tm, B, P + Ec)
    exp = torch.zeros(B, T, Ec, device=dev)
    exp[:, :Tm] = x[:, :, P:].transpose(0, 1)
    assert torch.equal(d_emb.view(B, T, Ec), exp)
    assert relerr(d_plan, x[:, :, :P].sum(0)) < 1e-6
    dx_pr, dS, dgin = rnd(B * T, D, seed=2).to(dev), rnd(B, 2 * Ec, seed=3).to(dev), rnd(B, Ec, seed=4).to(dev)
    f_dout = [torch.full((B * T, 32), float("nan"), device=dev) for _ in range(ncam)]
    ops.call("tacorl_plmp_demb_finish", ops.ptr(d_emb), ops.ptr(dx_pr), D, D_in, ops.ptr(dS), 2 * Ec, ops.ptr(dgin), ops.ptr_array(f_dout),
             ncam, B, T, Ec, ops.stream())
    torch.cuda.synchronize()
    want = exp + dx_pr.view(B, T, D)[:, :, :D_in]
    want[:, 0] = want[:, 0] + dS[:, :Ec]
    want[:, T - 1] = want[:, T - 1] + dgin
    assert torch.equal(d_emb.view(B, T, Ec), want)
    for j in range(ncam):
        assert torch.equal(f_dout[j], want.view(B * T, Ec)[:, 32 * j: 32 * j + 32])


def test_logistic_mixture_lazy_finish():
    """A logging-only loss may leave its per-block partial sums on the device (loss_out = NULL) and have them summed later:
    tacorl_logistic_mixture_finish gives the very two floats the one-call form writes."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    B, T, Da, K = 9, 7, 6, 10
    Tm, ldh = T - 1, 192
    heads = rnd(Tm * B, ldh, seed=5).to(dev)
    actions = (rnd(B, T, Da + 1, seed=6).clamp(-1, 1)).to(dev)
    actions[..., -1] = torch.where(actions[..., -1] >= 0, 1.0, -1.0)
    nb = _lib.lib().tacorl_logistic_mixture_ws_bytes(B, Tm, Da)
    outs = []
    for lazy in (False, True):
        ws = torch.zeros(max(256, nb), dtype=torch.uint8, device=dev)
        out = torch.full((2,), float("nan"), device=dev)
        ops.call("tacorl_logistic_mixture_loss", ops.ptr(heads), ldh, ops.ptr(actions), None, None if lazy else ops.ptr(out), B, T, Tm, Da, K,
                 10, 0.0095, 1.0, ops.ptr(ws), ws.numel(), ops.stream())
        if lazy:
            torch.cuda.synchronize()
            assert torch.isnan(out).all()  # nothing written yet
            ops.call("tacorl_logistic_mixture_finish", ops.ptr(ws), ws.numel(), B, Tm, Da, ops.ptr(out), ops.stream())
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_rnn_bptt_step_and_transpose():
    """BPTT step through the ring GEMM: (x Wt^T + addend) * [mask > 0] with Wt from the transpose kernel."""
    from tacorl_amd import ops

    dev = _dev()
    M, H = 256, 512
    dz = rnd(M, H, seed=1).to(torch.bfloat16)
    W = rnd(H, H, seed=2) / math.sqrt(H)
    add, hprev = rnd(M, H, seed=3), rnd(M, H, seed=4)
    ref = (dz.float() @ W.to(torch.bfloat16).float() + add) * (hprev > 0)
    Wd, wt = W.to(dev), torch.zeros(H, H, device=dev, dtype=torch.bfloat16)
    ops.call("tacorl_transpose_to_bf16", ops.ptr(Wd), ops.ptr(wt), H, H, ops.stream())
    y = torch.full((M, H), float("nan"), device=dev)
    yb = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
    dzd, addd, hd = dz.to(dev), add.to(dev), hprev.to(dev)
    ops.call("tacorl_rnn_linear_bwd_step", ops.ptr(dzd), ops.ptr(wt), ops.ptr(addd), H, ops.ptr(hd), ops.ptr(y), ops.ptr(yb),
             M, H, H, ops.stream())
    torch.cuda.synchronize()
    assert torch.equal(wt.cpu(), W.t().contiguous().to(torch.bfloat16))
    assert relerr(y, ref) < 1e-5, relerr(y, ref)
    assert torch.equal(yb, y.to(torch.bfloat16))


@pytest.mark.parametrize("compute,tol", [(0, TOL_F32), (1, TOL_BF16)])
@pytest.mark.parametrize("M,O,I,masked", [(512, 2048, 32, False), (256, 4096, 32, False), (300, 96, 32, False), (512, 2048, 32, True)])
def test_linear_dgrad_splitk(compute, tol, M, O, I, masked):
    """dX = dZ W (+ addend) (* act'(src)): the split-reduction entry vs the single-pass one and vs torch; with a mask
    source (or a short reduction) it must fall back to the single pass and still be right."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    dz, w, add, src = rnd(M, O, seed=1), rnd(O, I, seed=2) / math.sqrt(O), rnd(M, I, seed=3), rnd(M, I, seed=4)
    ref = dz @ w + add
    if masked:
        ref = ref * (src > 0)
    dzd, wd, addd, srcd = dz.to(dev), w.to(dev), add.to(dev), src.to(dev)
    outs = []
    for split in (False, True):
        out = torch.full((M, I), float("nan"), device=dev)
        args = [1, ops.ptr_array([dzd]), O, ops.ptr_array([wd]), ops.ptr_array([out]), I,
                ops.ptr_array([srcd]) if masked else None, I, 1 if masked else 0, ops.ptr_array([addd]), I, ops.int_array([M]), O, I,
                compute]
        if split:
            nb = _lib.lib().tacorl_linear_dgrad_ws_bytes(1, ops.int_array([M]), O, I)
            assert (nb > 0) == (O >= 512)
            ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
            ops.call("tacorl_linear_dgrad_splitk", *args, ops.ptr(ws), ws.numel(), ops.stream())
        else:
            ops.call("tacorl_linear_dgrad", *args, ops.stream())
        torch.cuda.synchronize()
        assert relerr(out, ref) < tol, (split, relerr(out, ref))
        outs.append(out)
    assert relerr(outs[1], outs[0]) < (1e-6 if compute == 0 else 1e-5)


@pytest.mark.parametrize("R,M,N,ld", [(64, 128, 128, 128), (192, 256, 384, 512), (1024, 512, 256, 512)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_rnn_wgrad(R, M, N, ld, accumulate):
    """dW = dz^T x, db = column sums of dz from bf16 row-major operands (transposing LDS reads, LDS-DMA ring) vs fp32
    torch on the same bf16 values: only the fp32 summation order differs."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    assert _lib.lib().tacorl_rnn_wgrad_supported(R, M, N) == 1 and _lib.lib().tacorl_rnn_wgrad_supported(R + 32, M, N) == 0
    dz, x = rnd(R, ld, seed=1).to(torch.bfloat16), rnd(R, ld, seed=2).to(torch.bfloat16)
    ref_w = dz[:, :M].float().t() @ x[:, :N].float()
    ref_b = dz[:, :M].float().sum(0)
    base = 0.5 if accumulate else float("nan")
    dw, db = torch.full((M, N), base, device=dev), torch.full((M,), base, device=dev)
    dzd, xd = dz.to(dev), x.to(dev)
    ops.call("tacorl_rnn_wgrad", ops.ptr(dzd), ld, ops.ptr(xd), ld, R, M, N, ops.ptr(dw), ops.ptr(db), int(accumulate), ops.stream())
    dw2 = torch.full((M, N), base, device=dev)
    ops.call("tacorl_rnn_wgrad", ops.ptr(dzd), ld, ops.ptr(xd), ld, R, M, N, ops.ptr(dw2), None, int(accumulate), ops.stream())
    torch.cuda.synchronize()
    off = 0.5 if accumulate else 0.0
    assert relerr(dw - off, ref_w) < 1e-5, relerr(dw - off, ref_w)
    assert relerr(db - off, ref_b) < 1e-5, relerr(db - off, ref_b)
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize("rows,M,accumulate", [((192, 192, 256), 256, False), ((3584, 3584, 3840, 3840), 2048, False), ((128,), 128, True)])
def test_rnn_wgrad_batch(rows, M, accumulate):
    """Several square weight gradients with different row counts in ONE launch (the RNN's W_hh_l / W_ih_l behind the BPTT):
    bit-identical to one tacorl_rnn_wgrad launch per matrix (same kernel body, same summation order); > 4 problems, a bad
    row count and misaligned pointers are refused."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    n = len(rows)
    dz = [rnd(r, M, seed=10 + i).to(torch.bfloat16).to(dev) for i, r in enumerate(rows)]
    x = [rnd(r, M, seed=20 + i).to(torch.bfloat16).to(dev) for i, r in enumerate(rows)]
    base = 0.5 if accumulate else float("nan")
    mk = lambda *s: torch.full(s, base, device=dev)  # noqa: E731
    dw, db = [mk(M, M) for _ in rows], [mk(M) if i % 2 else None for i in range(n)]
    dw1, db1 = [mk(M, M) for _ in rows], [mk(M) if i % 2 else None for i in range(n)]
    args = lambda R, W, Bs: (n, ops.ptr_array(dz), M, ops.ptr_array(x), M, ops.int_array(list(R)), M, M, ops.ptr_array(W),  # noqa: E731
                             ops.ptr_array(Bs), int(accumulate), ops.stream())
    ops.call("tacorl_rnn_wgrad_batch", *args(rows, dw, db))
    for i, r in enumerate(rows):
        ops.call("tacorl_rnn_wgrad", ops.ptr(dz[i]), M, ops.ptr(x[i]), M, r, M, M, ops.ptr(dw1[i]), ops.ptr(db1[i]) if db1[i] is not None else None,
                 int(accumulate), ops.stream())
    torch.cuda.synchronize()
    off = 0.5 if accumulate else 0.0
    for i in range(n):
        assert torch.equal(dw[i], dw1[i])
        assert relerr(dw[i] - off, dz[i].float().t() @ x[i].float()) < 1e-5
        if db[i] is not None:
            assert torch.equal(db[i], db1[i])
    f = _lib.lib().tacorl_rnn_wgrad_batch
    assert f(*args(tuple(r + 32 for r in rows), dw, db)) != 0
    assert f(5, *args(rows, dw, db)[1:]) != 0 and f(0, *args(rows, dw, db)[1:]) != 0


@pytest.mark.parametrize("R,slabs,rows,N,accumulate", [(3840, 6, 182, 2048, False), (512, 8, 182, 2048, True), (128, 2, 100, 256, False),
                                                        (64, 1, 128, 128, False)])
def test_rnn_wgrad_slabs(R, slabs, rows, N, accumulate):
    """dW[rows][N] = dz^T x and db = column sums for a FEW output rows (the action decoder's 182 x 2048 heads): dz K-padded to
    Mp = 128-multiple columns (zeros beyond `rows`), the reduction rows cut into `slabs` ranges that run side by side and
    are summed in slab order - vs fp32 torch on the same bf16 values; rows beyond `rows` of the destination stay untouched."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    Mp = (rows + 127) // 128 * 128
    dz = torch.zeros(R, Mp)
    dz[:, :rows] = rnd(R, rows, seed=1)
    dz, x = dz.to(torch.bfloat16), rnd(R, N, seed=2).to(torch.bfloat16)
    ref_w, ref_b = dz[:, :rows].float().t() @ x.float(), dz[:, :rows].float().sum(0)
    base = 0.25 if accumulate else float("nan")
    dw, db = torch.full((rows + 3, N), base, device=dev), torch.full((rows + 3,), base, device=dev)
    dzd, xd = dz.to(dev), x.to(dev)
    nb = _lib.lib().tacorl_rnn_wgrad_slabs_ws_bytes(slabs, Mp, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    ops.call("tacorl_rnn_wgrad_slabs", ops.ptr(dzd), Mp, ops.ptr(xd), N, R, Mp, N, rows, slabs, ops.ptr(dw), ops.ptr(db), int(accumulate),
             ops.ptr(ws), ws.numel(), ops.stream())
    torch.cuda.synchronize()
    off = 0.25 if accumulate else 0.0
    assert relerr(dw[:rows] - off, ref_w) < 1e-5, relerr(dw[:rows] - off, ref_w)
    assert relerr(db[:rows] - off, ref_b) < 1e-5, relerr(db[:rows] - off, ref_b)
    tail_w, tail_b = dw[rows:], db[rows:]
    assert (torch.isnan(tail_w).all() and torch.isnan(tail_b).all()) if not accumulate else (bool((tail_w == base).all()) and bool((tail_b == base).all()))
    # row ranges that do not divide into 64-row stages are refused
    rc = _lib.lib().tacorl_rnn_wgrad_slabs(ops.ptr(dzd), Mp, ops.ptr(xd), N, R, Mp, N, rows, slabs + (7 if R % (64 * (slabs + 7)) else 1), ops.ptr(dw),
                                           ops.ptr(db), 0, ops.ptr(ws), ws.numel(), ops.stream())
    assert rc != 0


@pytest.mark.parametrize("R", [100, 4096])
def test_add_layernorm_fwd_bwd(R):
    """y = LayerNorm(x + res) forward and backward (dv, dw, db) vs torch autograd; R = 4096 exercises the
    two-level column sums of the weight / bias gradients."""
    from tacorl_amd import _lib, ops

    dev = _dev()
    D = 32
    x, res = rnd(R, D, seed=1).requires_grad_(True), rnd(R, D, seed=2)
    w, b = (1 + 0.1 * rnd(D, seed=3)).requires_grad_(True), (0.1 * rnd(D, seed=4)).requires_grad_(True)
    y = F.layer_norm(x + res, (D,), w, b, 1e-5)
    dy = rnd(R, D, seed=5)
    (y * dy).sum().backward()
    xd, rd, wd, bd, dyd = (t.detach().to(dev) for t in (x, res, w, b, dy))
    yo, stats = torch.empty(R, D, device=dev), torch.empty(R, 2, device=dev)
    ops.call("tacorl_add_layernorm_fwd", ops.ptr(xd), ops.ptr(rd), ops.ptr(wd), ops.ptr(bd), ops.ptr(yo), ops.ptr(stats), R, D,
             1e-5, ops.stream())
    dv, dw, db = torch.empty(R, D, device=dev), torch.empty(D, device=dev), torch.empty(D, device=dev)
    ws = torch.empty(_lib.lib().tacorl_add_layernorm_bwd_ws_bytes(R, D), dtype=torch.uint8, device=dev)
    ops.call("tacorl_add_layernorm_bwd", ops.ptr(dyd), ops.ptr(xd), ops.ptr(rd), ops.ptr(wd), ops.ptr(stats), ops.ptr(dv),
             ops.ptr(dw), ops.ptr(db), R, D, 0, ops.ptr(ws), ws.numel(), ops.stream())
    torch.cuda.synchronize()
    assert relerr(yo, y) < 1e-5
    assert relerr(dv, x.grad) < 1e-4
    assert relerr(dw, w.grad) < 1e-4 and relerr(db, b.grad) < 1e-4


@pytest.mark.parametrize("T,D", [(16, 32), (32, 32), (32, 64), (16, 64)])
def test_plan_recognition_fused_sample_and_frozen_cache(T, D):
    """(T = 32: the real-world configuration's window - two 16-row tiles per sequence in the launch, round 4.)
    The in-launch posterior head + plan sample (fc -> mean_fc composed into one affine map) against the
    two-GEMM head + tacorl_pr_sample of the same module, and the frozen-weights cache: the weight-only
    preparation is skipped while the parameter block's version counter stands still and re-issued after a
    torch in-place update (what load_state_dict does)."""
    from tacorl_amd._lib import call, ptr, stream as ops_stream
    from tacorl_amd.networks.plan_recognition import PlanRecognition

    dev = _dev()
    B, A = 64, (16 if D == 32 else 32)  # (d_model 64 = two cameras, latent 32: the real-world configuration)
    pr = PlanRecognition(state_dim=D, latent_plan_dim=A, device=dev, num_heads=8, num_layers=2, encoder_hidden_size=2048,
                         fc_hidden_size=4096, max_position_embeddings=T, trainable=False)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for k, v in pr.blk.views.items():
            if k.endswith("weight") and v.dim() == 2:
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / math.sqrt(v.shape[1]))
            elif "norm" in k and k.endswith("weight"):
                v.copy_(1 + 0.1 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
    emb, eps = rnd(B * T, D, seed=3).to(dev), rnd(B, A, seed=4).to(dev)

    def reference():
        head = pr.forward(emb, D, B, T, 1, inference=True).clone()  # fused encoder, two bf16 GEMMs for the head
        plan = torch.zeros(B, A, device=dev)
        call("tacorl_pr_sample", ptr(head), ptr(eps), ptr(plan), None, None, B, A, float(pr.min_std), ops_stream())
        return head, plan

    def fused():
        plan = torch.zeros(B, A, device=dev)
        head = pr.forward(emb, D, B, T, 1, inference=True, sample=(eps, plan), frozen=True).clone()
        return head, plan

    h_ref, p_ref = reference()
    h_fus, p_fus = fused()
    torch.cuda.synchronize()
    assert relerr(h_fus, h_ref) < 2e-2 and relerr(p_fus, p_ref) < 2e-2, (relerr(h_fus, h_ref), relerr(p_fus, p_ref))
    ver = pr._prep_version
    h2, p2 = fused()  # cached preparation: same bits
    assert pr._prep_version == ver and torch.equal(h2, h_fus) and torch.equal(p2, p_fus)
    with torch.no_grad():
        pr.blk.views["mean_fc.bias"].add_(0.5)  # in-place update bumps the version counter
    h3, p3 = fused()
    torch.cuda.synchronize()
    assert pr._prep_version != ver
    assert relerr(h3[:, :A], h_fus[:, :A] + 0.5) < 1e-5, "stale composed head after a weight update"
    # the library's own optimiser writes through raw pointers: ops.adam_step announces it to the version counter
    from tacorl_amd import ops
    ver = pr._prep_version
    blk = pr.blk
    g_, m_, v_ = torch.ones_like(blk.param), torch.zeros_like(blk.param), torch.zeros_like(blk.param)
    ops.adam_step(blk.param, g_, m_, v_, 1e-2, 0.0, torch.zeros(1, dtype=torch.int32, device=dev))
    h4, _ = fused()
    torch.cuda.synchronize()
    assert pr._prep_version != ver and not torch.equal(h4, h3)


@pytest.mark.parametrize("T,D", [(16, 32), (32, 32), (32, 64), (16, 64)])
def test_plan_recognition_fused_encoder(T, D):
    """Single-launch frozen plan-recognition encoder (one workgroup per sequence; window 16 or 32) vs the per-kernel bf16
    path of the same module (same operand roundings) and vs an fp32 torch restatement at bf16 tolerance."""
    from tacorl_amd import ops
    from tacorl_amd.networks.plan_recognition import PlanRecognition

    dev = _dev()
    B, A = 37, (16 if D == 32 else 32)
    HD = D // 8
    pr = PlanRecognition(state_dim=D, latent_plan_dim=A, device=dev, num_heads=8, num_layers=2, encoder_hidden_size=2048,
                         fc_hidden_size=4096, max_position_embeddings=T, trainable=False)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, v in pr.blk.views.items():
            if k.endswith("weight") and v.dim() == 2:
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / math.sqrt(v.shape[1]))
            elif "norm" in k and k.endswith("weight"):
                v.copy_(1 + 0.1 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
    emb = rnd(B * T, D, seed=9).to(dev)
    h_ref = pr.forward(emb, D, B, T, 1).clone()
    pooled_ref = pr.pooled.clone()
    h_fus = pr.forward(emb, D, B, T, 1, inference=True).clone()
    torch.cuda.synchronize()
    assert getattr(pr, "_pb", None) is not None, "fused path not taken"
    assert relerr(pr.pooled, pooled_ref) < 1e-2, relerr(pr.pooled, pooled_ref)
    assert relerr(h_fus, h_ref) < 2e-2, relerr(h_fus, h_ref)
    # fp32 restatement (post-norm encoder layers, ReLU FFN, mean over time)
    P = {k: v.detach().cpu() for k, v in pr.blk.views.items()}
    x = emb.cpu().view(B, T, D) + P["position_embeddings.weight"][:T]
    for l in range(2):
        p = f"transformer_encoder.layers.{l}."
        qkv = x @ P[p + "self_attn.in_proj_weight"].t() + P[p + "self_attn.in_proj_bias"]
        q, k, v = (t.view(B, T, 8, HD).transpose(1, 2) for t in qkv.split(D, dim=-1))
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(HD), dim=-1) @ v
        att = att.transpose(1, 2).reshape(B, T, D) @ P[p + "self_attn.out_proj.weight"].t() + P[p + "self_attn.out_proj.bias"]
        x = F.layer_norm(x + att, (D,), P[p + "norm1.weight"], P[p + "norm1.bias"])
        ff = F.relu(x @ P[p + "linear1.weight"].t() + P[p + "linear1.bias"]) @ P[p + "linear2.weight"].t() + P[p + "linear2.bias"]
        x = F.layer_norm(x + ff, (D,), P[p + "norm2.weight"], P[p + "norm2.bias"])
    assert relerr(pr.pooled, x.mean(1)) < TOL_BF16, relerr(pr.pooled, x.mean(1))
    # the oracle's plan recognition with the MFMA's operand rounding: posterior mean and std of the fused launch
    from oracle import tacorl_oracle as O

    with O.operand_rounding(torch.bfloat16):
        mu, std = O.plan_recognition(P, "", emb.cpu().view(B, T, D), min_std=pr.min_std)
    e_mu = relerr(h_fus[:, :A], mu)
    e_sd = relerr(F.softplus(h_fus[:, A:]) + pr.min_std, std)
    assert e_mu < 5e-3 and e_sd < 5e-3, ("posterior vs oracle with bf16 operand rounding", e_mu, e_sd)


def test_plan_recognition_fused_train_forward_window32():
    """Window 32: the train-mode forward as one launch (two row tiles per sequence) - every saved tensor against the per-op
    forward's, and the per-op backward (the one-launch backward exists for window 16 only) run on either set of saves."""
    from tacorl_amd.networks.plan_recognition import PlanRecognition

    dev = _dev()
    B, T, D, A = 21, 32, 32, 32
    pr = PlanRecognition(state_dim=D, latent_plan_dim=A, device=dev, num_heads=8, num_layers=2, encoder_hidden_size=2048,
                         fc_hidden_size=4096, max_position_embeddings=T)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for k, v in pr.blk.views.items():
            if k.endswith("weight") and v.dim() == 2:
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / math.sqrt(v.shape[1]))
            elif "norm" in k and k.endswith("weight"):
                v.copy_(1 + 0.1 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
    emb, d_head = rnd(B * T, D, seed=19).to(dev), rnd(B, 2 * A, seed=20).to(dev)
    res = {}
    for fused in (False, True):
        pr.fused_train = fused
        for t in pr.x + pr.qkv + pr.att + pr.proj + pr.ff1 + pr.ff2 + pr.stats if pr._shape else []:
            t.fill_(float("nan"))
        head = pr.forward(emb, D, B, T, 1, train=True).clone()
        sv = [t.clone() for t in pr.x[: 2 * pr.L] + pr.qkv + pr.att + pr.proj + pr.ff1 + pr.ff2 + pr.stats]
        pr.blk.grad.zero_()
        dx = pr.backward(d_head, B, T, 1).clone()
        torch.cuda.synchronize()
        res[fused] = (head, sv, dx, pr.blk.grad.clone())
    assert pr._fused_saved == (B, T), "fused train forward not taken at T = 32"
    (h0, s0, dx0, g0), (h1, s1, dx1, g1) = res[False], res[True]
    for k, (a, b) in enumerate(zip(s0, s1)):
        assert torch.isfinite(b).all() and relerr(b, a) < 5e-3, (k, relerr(b, a))
    assert relerr(h1, h0) < 5e-3 and relerr(dx1, dx0) < 1e-2 and relerr(g1, g0) < 1e-2, (relerr(h1, h0), relerr(dx1, dx0), relerr(g1, g0))
    # composed head + plan sample inside the launch (prepare_inference() issued), per-op backward: fc_out is formed there
    eps, plan = rnd(B, A, seed=21).to(dev), torch.full((B, A), float("nan"), device=dev)
    pr.prepare_inference()
    pr.fc_out.fill_(float("nan"))
    hc = pr.forward(emb, D, B, T, 1, train=True, prepared=True, sample=(eps, plan)).clone()
    assert pr._composed and torch.isfinite(plan).all()
    pr.blk.grad.zero_()
    dxc = pr.backward(d_head, B, T, 1).clone()
    torch.cuda.synchronize()
    gc = pr.blk.grad.clone()
    assert relerr(hc, h0) < 5e-3 and relerr(dxc, dx0) < 1e-2 and torch.isfinite(gc).all() and relerr(gc, g0) < 1e-2, (relerr(hc, h0), relerr(dxc, dx0), relerr(gc, g0))


def test_plan_recognition_fused_train_forward():
    """Train-mode forward of the plan recognition as ONE launch (bf16, no dropout): every tensor it saves for the per-op
    backward against what the per-op forward saves (same operand roundings; the FFN's fp32 summation order differs), the
    posterior against the oracle with bf16 operand rounding, and the backward run on either set of saves gives the same
    gradients."""
    from oracle import tacorl_oracle as O
    from tacorl_amd import ops
    from tacorl_amd.networks.plan_recognition import PlanRecognition

    dev = _dev()
    B, T, D, A = 37, 16, 32, 16
    pr = PlanRecognition(state_dim=D, latent_plan_dim=A, device=dev, num_heads=8, num_layers=2, encoder_hidden_size=2048,
                         fc_hidden_size=4096, max_position_embeddings=T)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for k, v in pr.blk.views.items():
            if k.endswith("weight") and v.dim() == 2:
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / math.sqrt(v.shape[1]))
            elif "norm" in k and k.endswith("weight"):
                v.copy_(1 + 0.1 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
    emb = rnd(B * T, D, seed=19).to(dev)
    d_head = rnd(B, 2 * A, seed=20).to(dev)
    saves = lambda: {"x": [t.clone() for t in pr.x[: 2 * pr.L]], "qkv": [t.clone() for t in pr.qkv],  # noqa: E731
                     "att": [t.clone() for t in pr.att], "proj": [t.clone() for t in pr.proj],
                     "ff1": [t.clone() for t in pr.ff1], "ff2": [t.clone() for t in pr.ff2],
                     "stats": [t.clone() for t in pr.stats]}
    res = {}
    for fused, fused_bwd in ((False, False), (True, False), (True, True)):
        pr.fused_train, pr.fused_backward = fused, fused_bwd
        for t in pr.x + pr.qkv + pr.att + pr.proj + pr.ff1 + pr.ff2 + pr.stats if pr._shape else []:
            t.fill_(float("nan"))
        head = pr.forward(emb, D, B, T, 1, train=True).clone()
        sv = saves()
        pr.blk.grad.zero_()
        if fused_bwd:  # (every parameter's gradient must be WRITTEN by this path; the block's padding stays 0)
            for v in pr.blk.grad_views.values():
                v.fill_(float("nan"))
        if getattr(pr, "_bshape", None):
            for t in [pr.dx] + pr.dv + pr.dv1 + pr.d_ff1 + pr.d_qkv:
                t.fill_(float("nan"))
        dx = pr.backward(d_head, B, T, 1).clone()
        torch.cuda.synchronize()
        dz = {"dv": [t.clone() for t in pr.dv], "dv1": [t.clone() for t in pr.dv1], "d_ff1": [t.clone() for t in pr.d_ff1],
              "d_qkv": [t.clone() for t in pr.d_qkv]}
        res[(fused, fused_bwd)] = (head, sv, dx, pr.blk.grad.clone(), dz)
    # the input-gradient chain as ONE launch (round 4) against the per-op chain on the same saved tensors: every dZ operand
    # it hands to the weight-gradient GEMMs, the input gradient and all parameter gradients (LayerNorm's included)
    (_, _, dxp, gp, zp), (_, _, dxf, gf, zf) = res[(True, False)], res[(True, True)]
    bad = [k for k, (o, _, n) in pr.blk.off.items()  # ("layernorm.*": declared by the reference, never used in forward)
           if not k.startswith("layernorm.") and not torch.isfinite(gf[o: o + n]).all()]
    assert not bad and torch.isfinite(dxf).all(), ("not written / not finite", bad)
    for k in zp:
        for l, (a, b) in enumerate(zip(zp[k], zf[k])):
            assert torch.isfinite(b).all(), (k, l)
            assert relerr(b, a) < 6e-3, ("fused backward", k, l, relerr(b, a))
    assert relerr(dxf, dxp) < 6e-3, relerr(dxf, dxp)
    for name, (o, _, n) in pr.blk.off.items():
        if name.startswith("layernorm."):
            gf[o: o + n] = 0
            continue
        assert relerr(gf[o: o + n], gp[o: o + n]) < 1e-2, ("fused backward", name, relerr(gf[o: o + n], gp[o: o + n]))
    # composed posterior head (prepare_inference() issued: head = Wc pooled + bc and the plan sample inside the forward launch,
    # d_pool = d_head Wc inside the backward launch, fc_out / d_fc only for the two weight gradients)
    eps, plan = rnd(B, A, seed=21).to(dev), torch.full((B, A), float("nan"), device=dev)
    pr.fused_train = pr.fused_backward = True
    pr.prepare_inference()
    headc = pr.forward(emb, D, B, T, 1, train=True, prepared=True, sample=(eps, plan)).clone()
    assert pr._composed
    pr.blk.grad.zero_()
    for v in pr.blk.grad_views.values():
        v.fill_(float("nan"))
    dxc = pr.backward(d_head, B, T, 1).clone()
    torch.cuda.synchronize()
    gc = pr.blk.grad.clone()
    assert relerr(headc, res[(True, True)][0]) < 5e-3, relerr(headc, res[(True, True)][0])
    sd = F.softplus(headc[:, A:]) + pr.min_std
    assert relerr(plan, torch.tanh(headc[:, :A] + eps * sd)) < 1e-5
    assert relerr(dxc, dxf) < 6e-3, relerr(dxc, dxf)
    for name, (o, _, n) in pr.blk.off.items():
        if not name.startswith("layernorm."):
            assert torch.isfinite(gc[o: o + n]).all() and relerr(gc[o: o + n], gp[o: o + n]) < 1e-2, ("composed head", name)
    (h0, s0, dx0, g0, _), (h1, s1, dx1, g1, _) = res[(False, False)], res[(True, True)]
    for k in s0:
        for l, (a, b) in enumerate(zip(s0[k], s1[k])):
            assert torch.isfinite(b).all(), (k, l)
            assert relerr(b, a) < 5e-3, (k, l, relerr(b, a))
    assert relerr(h1, h0) < 5e-3, relerr(h1, h0)
    assert relerr(dx1, dx0) < 1e-2 and relerr(g1, g0) < 1e-2, (relerr(dx1, dx0), relerr(g1, g0))
    P = {k: v.detach().cpu() for k, v in pr.blk.views.items()}
    with O.operand_rounding(torch.bfloat16):
        mu, std = O.plan_recognition(P, "", emb.cpu().view(B, T, D), min_std=pr.min_std)
    e_mu = relerr(h1[:, :A], mu)
    e_sd = relerr(F.softplus(h1[:, A:]) + pr.min_std, std)
    assert e_mu < 5e-3 and e_sd < 5e-3, ("posterior vs oracle with bf16 operand rounding", e_mu, e_sd)


@pytest.mark.parametrize("dt", [torch.float32, torch.int64, torch.int32, torch.uint8, torch.bool])
def test_stage_transition_dtypes(dt):
    """reward = done = float(disp == 1) for every disp dtype a dataloader may deliver, plus the action-window copy
    (TACORL.get_rl_batch, reference tacorl.py:142-179), one launch."""
    from tacorl_amd._lib import call, ptr, stream

    dev = _dev()
    B = 300
    g = torch.Generator().manual_seed(3)
    base = torch.randint(-1, 4, (B,), generator=g)
    disp = (base == 1).to(dev) if dt == torch.bool else base.clamp_min(0).to(dt).to(dev) if dt == torch.uint8 else base.to(dt).to(dev)
    acts = rnd(B, 16, 7, seed=1).to(dev)
    reward, done, out = torch.zeros(B, device=dev), torch.zeros(B, device=dev), torch.zeros(B, 16, 7, device=dev)
    code = {torch.float32: 0, torch.int64: 1, torch.int32: 2, torch.uint8: 3, torch.bool: 3}[dt]
    call("tacorl_stage_transition", ptr(disp), code, ptr(reward), ptr(done), B, ptr(acts), ptr(out), acts.numel(), stream())
    torch.cuda.synchronize()
    ref = (disp == 1).float()
    assert torch.equal(reward, ref) and torch.equal(done, ref) and torch.equal(out, acts)


@pytest.mark.parametrize("B,cols,reps,ld_in,ld_out", [(256, 64, 13, 72, 64), (96, 71, 97, 80, 71), (5, 3, 2, 4, 3)])
def test_reduce_rows_mod_batch(B, cols, reps, ld_in, ld_out):
    """out[b][c] = sum_j in[j*B + b][c] for several (in, out) pairs in one launch.  The four waves of a workgroup take
    j = w, w + 4, ... each and meet in wave order: bit-exact against the same order in torch (and within fp32 rounding
    of the plain sum)."""
    from tacorl_amd import ops
    from tacorl_amd._lib import call, stream

    dev = _dev()
    ins = [rnd(reps * B, ld_in, seed=5 + k).to(dev) for k in range(2)]
    outs = [torch.zeros(B, ld_out, device=dev) for _ in range(2)]
    call("tacorl_reduce_rows_mod_batch", 2, ops.ptr_array(ins), ld_in, ops.ptr_array(outs), ld_out, B, cols, reps, stream())
    single = torch.zeros(B, ld_out, device=dev)
    call("tacorl_reduce_rows_mod", ops.ptr(ins[0]), ld_in, ops.ptr(single), ld_out, B, cols, reps, stream())
    torch.cuda.synchronize()
    for x, o in zip(ins, outs):
        part = []
        for w in range(4):
            acc = torch.zeros(B, cols, device=dev)
            for j in range(w, reps, 4):
                acc = acc + x[j * B:(j + 1) * B, :cols]
            part.append(acc)
        ref = ((part[0] + part[1]) + part[2]) + part[3]
        assert torch.equal(o[:, :cols], ref)
        plain = x.view(reps, B, ld_in)[:, :, :cols].double().sum(0)
        assert relerr(o[:, :cols], plain) < 1e-6
    assert torch.equal(single, outs[0])


@pytest.mark.parametrize("flag,dt", [(1, torch.bfloat16), (0, torch.float32)])
def test_pack_images_u8(flag, dt):
    """uint8 HWC frames -> normalised NHWC (ToTensor x/255, Normalize (t-0.5)/0.5 in fp32, as the reference's CPU
    transform pipeline computes them), strided jobs included: bit-exact."""
    from tacorl_amd import ops

    dev = _dev()
    n, T, H, W = 6, 4, 44, 60
    g = torch.Generator().manual_seed(2)
    x = torch.randint(0, 256, (n, T, H, W, 3), dtype=torch.uint8, generator=g)
    ref = ((x.float().div(255) - 0.5) / 0.5).to(dt)
    xd = x.to(dev)
    img = H * W * 3
    all_ = torch.zeros(n * T, H, W, 3, device=dev, dtype=dt)
    first = torch.zeros(n, H, W, 3, device=dev, dtype=dt)
    ops.pack_images_u8_batch([(xd.data_ptr(), img, all_.data_ptr(), n * T), (xd.data_ptr(), T * img, first.data_ptr(), n)],
                             flag, H, W)
    torch.cuda.synchronize()
    assert torch.equal(all_.cpu().view(n, T, H, W, 3), ref)
    assert torch.equal(first.cpu(), ref[:, 0])


@pytest.mark.parametrize("H", [64, 192, 256])
def test_action_decoder_frozen_bf16_hidden_sizes(H):
    """bf16, frozen decoder (TACORL's logging-only pass, validation): hidden sizes the ring-GEMM path does not take (64,
    192: H % 128 != 0) go through the generic per-step path, which reads x_seq - it must have been built for THIS batch
    (round-3 advisor finding: it was skipped whenever the fused input projection's own gate held).  The heads of the
    frozen pass must equal the non-frozen pass on the same inputs and sit at bf16 distance from the f32 path."""
    from tacorl_amd import ops
    from tacorl_amd.init import init_views_
    from tacorl_amd.networks.action_decoder import ActionDecoderLogistic

    dev = _dev()
    B, T, P, E = 8, 6, 16, 32
    torch.manual_seed(11)
    ad = ActionDecoderLogistic(dev, state_dim=E, latent_plan_dim=P, hidden_size=H, out_features=7, num_layers=2)
    init_views_(ad.blk.views, rnn_hidden=H)
    plan = rnd(B, P, seed=1).to(dev)
    emb = rnd(B * T, E, seed=2).to(dev)
    res = {}
    for tag, compute, frozen in (("f32", ops.F32, False), ("bf16", ops.BF16, False), ("bf16_frozen", ops.BF16, True)):
        ad._ensure(B, T - 1)
        ad.x_seq.fill_(float("nan"))  # a stale / unbuilt input must show
        ad._bf16_version = None
        ad.forward(plan, emb, E, B, T, T - 1, compute, frozen=frozen)
        torch.cuda.synchronize()
        res[tag] = ad.heads[:, :ad.NH].clone()
        assert torch.isfinite(res[tag]).all(), tag
    assert relerr(res["bf16_frozen"], res["bf16"]) < 1e-6
    assert relerr(res["bf16"], res["f32"]) < TOL_BF16


@pytest.mark.parametrize("B,T", [(16, 6), (256, 16)])
def test_action_decoder_heads_dgrad_ring(B, T):
    """bf16 backward of the action decoder: dH = d_heads W through the ring GEMM (d_heads and W^T as K-padded bf16 operands:
    tacorl_pad_to_bf16 / tacorl_transpose_pad_to_bf16, K = 182 -> 256) against the generic bf16 GEMM on the same
    operands - same bf16 products, fp32 accumulation in another order - and the two helper kernels against torch."""
    from tacorl_amd import ops
    from tacorl_amd.init import init_views_
    from tacorl_amd.networks.action_decoder import ActionDecoderLogistic

    dev = _dev()
    P, E, H = 16, 32, 256 if B < 100 else 2048
    torch.manual_seed(12)
    ad = ActionDecoderLogistic(dev, state_dim=E, latent_plan_dim=P, hidden_size=H, out_features=7, num_layers=2)
    init_views_(ad.blk.views, rnn_hidden=H)
    plan, emb = rnd(B, P, seed=1).to(dev), rnd(B * T, E, seed=2).to(dev)
    acts = rnd(B, T, 7, seed=3).clamp(-1, 1).to(dev)
    acts[..., 6] = torch.sign(acts[..., 6])
    loss = torch.zeros(1, device=dev)
    res = {}
    for ring in (True, False):  # (True: also the heads' weight gradient through the row-slabbed transposing-read kernel)
        ad.heads_dgrad_ring = ring
        ad.forward(plan, emb, E, B, T, T - 1, ops.BF16)
        ad.loss(acts, ops.ptr(loss), B, T, T - 1, want_grad=True)
        ad.blk.grad.zero_()
        ad.backward(B, T - 1, ops.BF16, need_input_grad=True)
        torch.cuda.synchronize()
        res[ring] = (ad.dH.clone(), ad.blk.grad.clone(), ad.dx_seq.clone())
    assert getattr(ad, "d_heads_b", None) is not None, "the ring path did not run"
    for a, b, name in zip(res[True], res[False], ("dH", "grad", "dx_seq")):
        assert torch.isfinite(a).all() and relerr(a, b) < 2e-5, (name, relerr(a, b))
    R, KP = B * (T - 1), ad.d_heads_b.shape[1]
    ref = torch.zeros(R, KP, device=dev)
    ref[:, : ad.NH] = ad.d_heads[:, : ad.NH]
    assert torch.equal(ad.d_heads_b[:R], ref.to(torch.bfloat16))
    assert ad.d_heads_b.shape[0] % 64 == 0 and not ad.d_heads_b[R:].any() and not ad.hb[-1][R:].any()  # (zero pad rows: slab operands)
    wt = torch.zeros(H, KP, device=dev)
    o = ad.blk.off["mean_fc.weight"][0]  # (the four heads' weights sit back to back: one NH x H matrix)
    wt[:, : ad.NH] = ad.blk.param[o: o + ad.NH * H].view(ad.NH, H).t()
    assert torch.equal(ad.headwt_b, wt.to(torch.bfloat16))


@pytest.mark.parametrize("dropout", [False, True])
@pytest.mark.parametrize("B", [5, 40])
def test_attention_fwd_bwd_t16(B, dropout):
    """Multi-head self-attention core of the plan recognition (T = 16, 8 heads of 4: the 16-lanes-per-(sequence, head) backward
    kernel) against torch autograd on the same q|k|v, with and without the keep mask on the attention probabilities."""
    from tacorl_amd import _lib
    from tacorl_amd._lib import call, ptr
    from tacorl_amd import ops

    dev = _dev()
    T, D, H = 16, 32, 8
    qkv = rnd(B * T, 3 * D, seed=31, scale=1.5)
    d_out = rnd(B * T, D, seed=32)
    keep = (torch.rand(B, H, T, T, generator=torch.Generator().manual_seed(33)) > 0.2) if dropout else None
    ks = 1.0 / 0.8
    q = qkv.clone().requires_grad_(True)
    qq, kk, vv = (t.reshape(B, T, H, 4).transpose(1, 2) for t in q.split(D, dim=-1))
    p = torch.softmax(qq @ kk.transpose(-1, -2) / 2.0, dim=-1)
    if dropout:
        p = p * keep.float() * ks
    out = (p @ vv).transpose(1, 2).reshape(B * T, D)
    (out * d_out).sum().backward()
    qd, dd = qkv.to(dev), d_out.to(dev)
    o = torch.full((B * T, D), float("nan"), device=dev)
    dq = torch.full((B * T, 3 * D), float("nan"), device=dev)
    if dropout:
        kd = keep.to(torch.uint8).to(dev).contiguous()
        call("tacorl_attention_dropout_fwd", ptr(qd), ptr(o), ptr(kd), ks, B, T, D, H, ops.stream())
        call("tacorl_attention_dropout_bwd", ptr(qd), ptr(dd), ptr(dq), ptr(kd), ks, B, T, D, H, ops.stream())
    else:
        call("tacorl_attention_fwd", ptr(qd), ptr(o), B, T, D, H, ops.stream())
        call("tacorl_attention_bwd", ptr(qd), ptr(dd), ptr(dq), B, T, D, H, ops.stream())
    torch.cuda.synchronize()
    assert relerr(o, out) < TOL_F32, relerr(o, out)
    assert torch.isfinite(dq).all() and relerr(dq, q.grad) < TOL_F32, relerr(dq, q.grad)


def test_plan_recognition_fused_backward_refuses_shapes_it_is_not_built_for():
    """the forward launch also takes d_model 64 / window 32, the backward launch is d_model 32, window 16 only
    (64-column LayerNorm partials, [32][*] transposes): a direct C-ABI call with another shape must return EINVAL before
    anything is launched, not run out of bounds.  All pointers valid-looking and 16-byte aligned so that only the shape
    check can refuse."""
    import ctypes as C

    from tacorl_amd import _lib, ops

    dev = _dev()
    buf = torch.zeros(1 << 16, device=dev)
    p = ops.ptr(buf)
    offs = (C.c_long * 40)(*([0] * 40))
    pa = _lib.ptr_array([buf] * 18)
    L = _lib.lib()
    assert L.tacorl_pr_encoder_fused_supported(64, 32, 8, 2048, 2) == 1
    for D, T in ((64, 32), (64, 16), (32, 32)):
        assert L.tacorl_pr_encoder_fused_train_supported(D, T, 8, 2048, 2) in (0, 1)
        rc = L.tacorl_pr_encoder_bwd_fused(p, offs, p, p, p, 32, p, pa, pa, pa, p, pa, 4, D, T, 8, 2048, 2, ops.stream())
        assert rc != 0, (D, T)
    torch.cuda.synchronize()
