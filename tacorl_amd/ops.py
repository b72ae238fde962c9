"""Thin torch-tensor wrappers over the C ABI (device memory + streams are torch's; the
arithmetic is libtacorl_hip.so's).  No function here computes anything itself."""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import ACT_NONE, ACT_RELU, ACT_SILU, BF16, F32, call, int_array, ptr, ptr_array, stream  # noqa: F401

import threading

_ws_cache = {}
_alloc_epoch = 0
# Held by a hipGraph capture (modules/common.py GraphMixin._run_segments: warm-up + capture) and by a feeder thread while it
# prepares a batch's device-side tables (data/replay.py prefetching): pinned / device allocations, event synchronisation or
# stream waits issued from another thread during a capture can fail it (hipErrorStreamCaptureUnsupported).  Re-entrant:
# a capture may nest helpers that take it again.
capture_lock = threading.RLock()


def note_alloc():
    """Every (re)allocation of a buffer that a captured hipGraph may hold a raw pointer to - step buffers
    of a new batch shape, a regrown workspace - bumps this epoch.  GraphMixin._run_segments stamps its
    captures with the epoch and drops any capture whose stamp is stale, so a graph is never replayed
    against freed memory (the reference DataLoader has no drop_last: B goes 256 -> r -> 256 every epoch)."""
    global _alloc_epoch
    _alloc_epoch += 1


def alloc_epoch():
    return _alloc_epoch


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag); its address is stable until it has to grow, and
    growing bumps the allocation epoch (captured graphs that point at the old buffer are dropped)."""
    key = (str(device), tag)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
        note_alloc()
    return t


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), (t.dtype, t.device, t.is_contiguous())
    return t


# ------------------------------------------------------------------------- layouts
def encoder_param_layout():
    off = (C.c_long * 11)()
    total = L.lib().tacorl_encoder_param_layout(off)
    return list(off), int(total)


def encoder_act_layout(n_img, H, W):
    off = (C.c_long * 5)()
    total = L.lib().tacorl_encoder_act_layout(int(n_img), H, W, off)
    if total < 0:
        raise L.TacorlHipError(f"encoder: image {H}x{W} too small for the conv stack")
    return list(off), int(total)


def mlp_param_layout(dims):
    n = len(dims) - 1
    w, b = (C.c_long * n)(), (C.c_long * n)()
    total = L.lib().tacorl_mlp_param_layout(n, int_array(dims), w, b)
    return list(w), list(b), int(total)


def mlp_act_layout(M, dims, acts):
    n = len(dims) - 1
    z, y = (C.c_long * n)(), (C.c_long * n)()
    total = L.lib().tacorl_mlp_act_layout(int(M), n, int_array(dims), int_array(acts), z, y)
    return list(z), list(y), int(total)


# ------------------------------------------------------------------------ primitives
def linear_fwd(xs, ws, bs, act=ACT_NONE, compute=F32, want_z=False):
    M = [x.shape[0] for x in xs]
    K, N = xs[0].shape[1], ws[0].shape[0]
    ys = [torch.empty(m, N, device=xs[0].device) for m in M]
    zs = [torch.empty(m, N, device=xs[0].device) for m in M] if want_z else None
    call("tacorl_linear_fwd", len(xs), ptr_array([_f32(x) for x in xs]), K, ptr_array(ws), ptr_array(bs),
         ptr_array(ys), ptr_array(zs) if zs else None, int_array(M), K, N, act, compute, stream())
    return (ys, zs) if want_z else ys


def conv2d_relu_fwd(xs, ws, bs, stride, compute=F32):
    """xs: NHWC (fp32 or bf16); ws: [CO][KH][KW][CI] fp32."""
    n = [x.shape[0] for x in xs]
    _, H, W, Ci = xs[0].shape
    CO, KH, KW, _ = ws[0].shape
    OH, OW = (H - KH) // stride + 1, (W - KW) // stride + 1
    ys = [torch.empty(k, OH, OW, CO, device=xs[0].device) for k in n]
    xd = BF16 if xs[0].dtype == torch.bfloat16 else F32
    call("tacorl_conv2d_relu_fwd", len(xs), ptr_array(xs), ptr_array(ws), ptr_array(bs), ptr_array(ys),
         int_array(n), H, W, Ci, KH, KW, stride, CO, xd, compute, stream())
    return ys


# --------------------------------------------------------------------------- encoder
def encoder_fwd(imgs, params, outs, acts, H, W, compute=F32):
    n = [i.shape[0] for i in imgs]
    xd = BF16 if imgs[0].dtype == torch.bfloat16 else F32
    call("tacorl_encoder_fwd", len(imgs), ptr_array(imgs), ptr_array(params), ptr_array(outs), ptr_array(acts),
         int_array(n), H, W, xd, compute, stream())


def encoder_bwd(imgs, params, acts, d_outs, grads, H, W, compute=F32, accumulate=False, fused=False):
    """fused=True: the per-image LDS-resident conv backward (bf16 images + bf16 MFMA + supported geometry)."""
    n = [i.shape[0] for i in imgs]
    xd = BF16 if imgs[0].dtype == torch.bfloat16 else F32
    if fused:
        assert xd == BF16 and compute == BF16
        nb = L.lib().tacorl_encoder_bwd_fused_ws_bytes(len(imgs), int_array(n), H, W)
        if nb == 0:
            raise L.TacorlHipError(f"encoder_bwd_fused: geometry {H}x{W} / {len(imgs)} problems not supported")
        ws = workspace(nb, imgs[0].device, "enc_bwd_fused")
        call("tacorl_encoder_bwd_fused", len(imgs), ptr_array(imgs), ptr_array(params), ptr_array(acts),
             ptr_array(d_outs), ptr_array(grads), int_array(n), H, W, int(accumulate), ptr(ws), ws.numel(), stream())
        return
    nb = L.lib().tacorl_encoder_bwd_ws_bytes(len(imgs), int_array(n), H, W)
    ws = workspace(nb, imgs[0].device, "enc_bwd")
    call("tacorl_encoder_bwd", len(imgs), ptr_array(imgs), ptr_array(params), ptr_array(acts), ptr_array(d_outs),
         ptr_array(grads), int_array(n), H, W, xd, compute, int(accumulate), ptr(ws), ws.numel(), stream())


# ------------------------------------------------------------------------------- MLP
def mlp_lean_ok(n, dims, ldx, ldo, ldd, compute):
    """Forward, input-gradient chain and weight gradients of this MLP site all run as the fused launches: the forward
    may then skip the hidden layers' outputs (mlp_fwd / mlp_bwd_fused_wgrad lean=True on BOTH)."""
    return compute == BF16 and bool(L.lib().tacorl_mlp_lean_supported(n, len(dims) - 1, int_array(dims), ldx, ldo, ldd))


def mlp_fwd(xs, ldx, params, acts_buf, M, dims, acts, compute=F32, params_bf16=None, lean=False):
    """params_bf16: bf16 copies of the parameter blocks -> the whole MLP runs as one launch when the
    shapes qualify (bf16 mode only); otherwise one GEMM launch per layer.
    lean: do not save hidden-layer outputs whose pre-activation is saved (see mlp_lean_ok)."""
    if (params_bf16 is not None and compute == BF16
            and L.lib().tacorl_mlp_fwd_fused_supported(len(xs), len(dims) - 1, int_array(dims), ldx)):
        call("tacorl_mlp_fwd_fused", len(xs), ptr_array(xs), ldx, ptr_array(params), ptr_array(params_bf16),
             ptr_array(acts_buf), int_array(M), len(dims) - 1, int_array(dims), int_array(acts), int(lean), stream())
        return
    assert not lean, "lean activations need the fused forward"
    call("tacorl_mlp_fwd", len(xs), ptr_array(xs), ldx, ptr_array(params), ptr_array(acts_buf), int_array(M),
         len(dims) - 1, int_array(dims), int_array(acts), compute, stream())


def mlp_fwd_gather_ok(M, dims, acts, ldx, compute, lean):
    return compute == BF16 and bool(L.lib().tacorl_mlp_fwd_fused_gather_supported(
        len(M), int_array(M), len(dims) - 1, int_array(dims), int_array(acts), ldx, int(lean)))


def mlp_fwd_gather(segs, x_out, ldx, params, params_bf16, acts_buf, M, dims, acts, lean=False):
    """The fused forward with gathered layer-0 inputs.  segs[p] = [(tensor, float offset, ld, first column, row modulo)],
    <= 4 per problem, in column order; x_out[p]: tensor that also receives the assembled fp32 rows, or None."""
    n = len(segs)
    sp, sl, sc, sm = [0] * (4 * n), [0] * (4 * n), [0] * (4 * n), [0] * (4 * n)
    for p, ss in enumerate(segs):
        assert 1 <= len(ss) <= 4
        for t, (ten, off, ld, c0, mod) in enumerate(ss):
            sp[4 * p + t], sl[4 * p + t], sc[4 * p + t], sm[4 * p + t] = ten.data_ptr() + 4 * off, ld, c0, mod
    call("tacorl_mlp_fwd_fused_gather", n, int_array([len(ss) for ss in segs]), (C.c_void_p * (4 * n))(*[v or None for v in sp]),
         int_array(sl), int_array(sc), int_array(sm), ptr_array(x_out), ldx, ptr_array(params), ptr_array(params_bf16),
         ptr_array(acts_buf), int_array(M), len(dims) - 1, int_array(dims), int_array(acts), int(lean), stream())


def mlp_bwd(xs, ldx, params, acts_buf, d_outs, ldo, grads, d_xs, ldd, M, dims, acts, compute=F32, accumulate=False,
            ws_tag="mlp_bwd"):
    """ws_tag: scratch buffer name - calls that may overlap on different streams need different tags."""
    nb = L.lib().tacorl_mlp_bwd_ws_bytes(len(xs), int_array(M), len(dims) - 1, int_array(dims))
    ws = workspace(nb, acts_buf[0].device, ws_tag)
    call("tacorl_mlp_bwd", len(xs), ptr_array(xs), ldx, ptr_array(params), ptr_array(acts_buf), ptr_array(d_outs), ldo,
         ptr_array(grads), ptr_array(d_xs) if d_xs is not None else None, ldd, int_array(M), len(dims) - 1,
         int_array(dims), int_array(acts), compute, int(accumulate), ptr(ws), ws.numel(), stream())


def mlp_bwd_fused_ok(n, dims, ldo, ldd, compute):
    return compute == BF16 and bool(L.lib().tacorl_mlp_bwd_fused_supported(n, len(dims) - 1, int_array(dims), ldo, ldd))


def mlp_bwd_fused_pack(params, M, dims, ws_tag, device):
    """Weight transposes for mlp_bwd_fused_dgrad(prepacked=True) with the same (M, dims, ws_tag)."""
    nb = L.lib().tacorl_mlp_bwd_fused_ws_bytes(len(params), int_array(M), len(dims) - 1, int_array(dims))
    ws = workspace(nb, device, ws_tag)
    call("tacorl_mlp_bwd_fused_pack", len(params), ptr_array(params), int_array(M), len(dims) - 1, int_array(dims), ptr(ws),
         ws.numel(), stream())


def mlp_bwd_fused_dgrad(params, acts_buf, d_outs, ldo, d_xs, ldd, M, dims, acts, ws_tag, prepacked=False, lean=False):
    """Input-gradient chain of the whole MLP in one launch; leaves every layer's dZ in workspace(ws_tag)
    for mlp_bwd_fused_wgrad (same ws_tag, same shapes, same `lean` as the forward and the weight gradients: at
    >= 16 384 rows a lean site hands dZ over as bf16 for the LDS-DMA weight-gradient kernel)."""
    nb = L.lib().tacorl_mlp_bwd_fused_ws_bytes(len(params), int_array(M), len(dims) - 1, int_array(dims))
    ws = workspace(nb, acts_buf[0].device, ws_tag)
    call("tacorl_mlp_bwd_fused_dgrad", len(params), ptr_array(params), ptr_array(acts_buf), ptr_array(d_outs), ldo,
         ptr_array(d_xs) if d_xs is not None else ptr_array([None] * len(params)), ldd, int_array(M), len(dims) - 1,
         int_array(dims), int_array(acts), int(bool(prepacked)) | (2 if lean else 0), ptr(ws), ws.numel(), stream())


def mlp_bwd_fused_wgrad(xs, ldx, acts_buf, d_outs, ldo, grads, M, dims, acts, ws_tag, accumulate=False, lean=False):
    nb = L.lib().tacorl_mlp_bwd_fused_ws_bytes(len(xs), int_array(M), len(dims) - 1, int_array(dims))
    ws = workspace(nb, acts_buf[0].device, ws_tag)
    call("tacorl_mlp_bwd_fused_wgrad", len(xs), ptr_array(xs), ldx, ptr_array(acts_buf), ptr_array(d_outs), ldo,
         ptr_array(grads), int_array(M), len(dims) - 1, int_array(dims), int_array(acts), int(accumulate), int(lean), ptr(ws),
         ws.numel(), stream())


# --------------------------------------------------------------------- data movement
def pack_images(src, img_pitch, src_nchw, dst, n, Cc, H, W, src_offset=0):
    dd = BF16 if dst.dtype == torch.bfloat16 else F32
    call("tacorl_pack_images", C.c_void_p(src.data_ptr() + 4 * src_offset), img_pitch, int(src_nchw), ptr(dst), dd,
         n, Cc, H, W, stream())


def pack_images_batch(jobs, dst_dtype_flag, H, W):
    """jobs: [(src_ptr, image_pitch_elems, dst_ptr, n_images)], NCHW fp32 (C = 3) -> NHWC, one launch
    (H*W % 4 == 0, 16-byte aligned pointers, pitches % 4 == 0)."""
    k = len(jobs)
    call("tacorl_pack_images_batch", k, (C.c_void_p * k)(*[j[0] for j in jobs]), (C.c_long * k)(*[j[1] for j in jobs]),
         (C.c_void_p * k)(*[j[2] for j in jobs]), int_array([j[3] for j in jobs]), dst_dtype_flag, H, W, stream())


def pack_images_u8_batch(jobs, dst_dtype_flag, H, W):
    """jobs: [(src_ptr, image_pitch_bytes, dst_ptr, n_images)], uint8 HWC frames -> normalised NHWC, one launch."""
    k = len(jobs)
    call("tacorl_pack_images_u8_batch", k, (C.c_void_p * k)(*[j[0] for j in jobs]), (C.c_long * k)(*[j[1] for j in jobs]),
         (C.c_void_p * k)(*[j[2] for j in jobs]), int_array([j[3] for j in jobs]), dst_dtype_flag, H, W, stream())


def pack_images_u8_aug_batch(jobs, dst_dtype_flag, H, W, pad):
    """jobs: [(src_ptr, image_pitch_bytes, dst_ptr, n_images, shift int32 (n,2) or None, jitter f32 (n,8) or None)]:
    uint8 HWC frames -> RandomShiftsAug -> /255 -> ColorJitter -> Normalize -> NHWC, one launch."""
    k = len(jobs)
    for j in jobs:
        for t, dt in ((j[4], torch.int32), (j[5], torch.float32)):
            assert t is None or (t.is_cuda and t.is_contiguous() and t.dtype == dt and t.shape[0] == j[3])
    call("tacorl_pack_images_u8_aug_batch", k, (C.c_void_p * k)(*[j[0] for j in jobs]), (C.c_long * k)(*[j[1] for j in jobs]),
         (C.c_void_p * k)(*[j[2] for j in jobs]), ptr_array([j[4] for j in jobs]), ptr_array([j[5] for j in jobs]),
         int_array([j[3] for j in jobs]), dst_dtype_flag, H, W, int(pad), stream())


def pack_images_u8_resize_aug_batch(jobs, dst_dtype_flag, src_hw, H, W, pad):
    """jobs: [(src_ptr, source_frame_bytes, dst_ptr, n_images, index_ptr or None, index_stride, shift or None, jitter or
    None)]: uint8 HWC source frames of size src_hw -> Resize(H, W) -> RandomShiftsAug(pad) -> /255 -> ColorJitter ->
    Normalize -> NHWC (the whole train pipeline of rl_train.yaml), one launch."""
    k = len(jobs)
    call("tacorl_pack_images_u8_resize_aug_gather_batch", k, (C.c_void_p * k)(*[j[0] for j in jobs]),
         (C.c_long * k)(*[j[1] for j in jobs]), (C.c_void_p * k)(*[j[4] for j in jobs]), int_array([j[5] for j in jobs]),
         (C.c_void_p * k)(*[j[2] for j in jobs]), ptr_array([j[6] for j in jobs]), ptr_array([j[7] for j in jobs]),
         int_array([j[3] for j in jobs]), dst_dtype_flag, int(src_hw[0]), int(src_hw[1]), H, W, int(pad), stream())


def pack_images_u8_gather_batch(jobs, dst_dtype_flag, H, W, pad=None):
    """jobs: [(dataset_ptr, frame_bytes, dst_ptr, n_images, index_ptr (device int64) or None, index_stride[, shift, jitter])]:
    image i = dataset frame index[i * stride], normalised (pad given: augmented, with the per-image tables) on the way
    into the NHWC image buffers - the replay gather and the pack in one pass."""
    k = len(jobs)
    base = [k, (C.c_void_p * k)(*[j[0] for j in jobs]), (C.c_long * k)(*[j[1] for j in jobs]),
            (C.c_void_p * k)(*[j[4] for j in jobs]), int_array([j[5] for j in jobs]), (C.c_void_p * k)(*[j[2] for j in jobs])]
    if pad is None:
        call("tacorl_pack_images_u8_gather_batch", *base, int_array([j[3] for j in jobs]), dst_dtype_flag, H, W, stream())
    else:
        call("tacorl_pack_images_u8_aug_gather_batch", *base, ptr_array([j[6] for j in jobs]), ptr_array([j[7] for j in jobs]),
             int_array([j[3] for j in jobs]), dst_dtype_flag, H, W, int(pad), stream())


def gather_frames_u8(frames, index, out):
    """out[i] = frames[index[i]] (uint8 frames resident in HBM, device int64 index)."""
    fb = frames[0].numel()
    assert frames.dtype == torch.uint8 and frames.is_contiguous() and index.dtype == torch.int64 and index.is_cuda
    assert out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == index.numel() * fb
    call("tacorl_gather_frames_u8", ptr(frames), fb, ptr(index), ptr(out), index.numel(), stream())
    return out


def _at(t, off):
    return C.c_void_p(t.data_ptr() + 4 * off)


# ---- tracing aid: device time marks between the launches of a step (off unless trace_marks() is called)
_marks = None


def trace_marks(device, slots=64):
    """Switch time marks on: every mark(name) call appends one 1-thread launch that stores the device clock.
    Returns a reader: reader() -> [(name, microseconds since the first mark)] of the last run / graph replay."""
    global _marks
    import torch

    buf = torch.zeros(slots, dtype=torch.int64, device=device)
    _marks = {"buf": buf, "names": []}

    def reader():
        t = buf.cpu().tolist()
        names = _marks["names"]
        t0 = min(t[: len(names)]) if names else 0
        return sorted(((n, (t[i] - t0) / 100.0) for i, n in enumerate(names)), key=lambda r: r[1])

    return reader


def mark(name):
    if _marks is None:
        return
    names = _marks["names"]
    if name not in names:
        names.append(name)
    call("tacorl_time_mark", ptr(_marks["buf"]), names.index(name), stream())


class prep_batch:
    """`with ops.prep_batch():` - the weight-only preparation launches issued inside (to_bf16 mirrors, the MLP backward
    sites' transposed weights, the encoders' FC-tail transposes) become ONE launch at the end of the block, on the
    stream current THERE (tacorl_prep_batch_begin / _end).  TACORL_PREP_BATCH=0: every launch on its own, as before."""
    enabled = os.environ.get("TACORL_PREP_BATCH", "1") == "1"

    def __enter__(self):
        self.on = prep_batch.enabled
        if self.on:
            call("tacorl_prep_batch_begin")
        return self

    def __exit__(self, et, ev, tb):
        if self.on:
            rc = L.lib().tacorl_prep_batch_end(stream())
            if et is None and rc != 0:
                raise RuntimeError(f"tacorl_prep_batch_end failed ({rc})")
        return False


class reduce_batch:
    """`with ops.reduce_batch():` - the slab reduces of the one-launch MLP weight gradients issued inside are summed by ONE
    launch at the end of the block (on the stream current there); the gradients are complete only after it.
    MEASURED SLOWER on the headline step, default off (same-process A/B, scratch/ab_step.py ops:reduce_batch.enabled:
    0.8201 -> 0.8276 ms/step): the five reduces run in launch gaps of the step's other branch where they stand; as one
    launch at the end of the backward they are 10 us of their own in front of the optimiser.  TACORL_REDUCE_BATCH=1 enables."""
    enabled = os.environ.get("TACORL_REDUCE_BATCH", "0") == "1"

    def __init__(self, auto=False):
        """auto: the caller's own judgement for this block (the actor-critic engine: many-row Q problems, where the reduces
        are large enough for one launch to win - C5 1.597 -> 1.586 ms/step); TACORL_REDUCE_BATCH=0 / 1 overrides it."""
        self.auto = bool(auto)

    def __enter__(self):
        env = os.environ.get("TACORL_REDUCE_BATCH")
        self.on = reduce_batch.enabled or (self.auto and env != "0")
        if self.on:
            call("tacorl_reduce_batch_begin")
        return self

    def __exit__(self, et, ev, tb):
        if self.on:
            rc = L.lib().tacorl_reduce_batch_end(stream())
            if et is None and rc != 0:
                raise RuntimeError(f"tacorl_reduce_batch_end failed ({rc})")
        return False


_copy_batch = None


class copy_batch:
    """`with ops.copy_batch():` turns the copy_cols calls inside into ONE launch (<= 32 per launch).
    The copies of a batch must be independent: none may read what another one writes."""

    def __enter__(self):
        global _copy_batch
        assert _copy_batch is None, "copy_batch does not nest"
        _copy_batch = []
        return self

    def __exit__(self, et, ev, tb):
        global _copy_batch
        items, _copy_batch = _copy_batch, None
        if et is not None:
            return False
        for i in range(0, len(items), 32):
            ch = items[i:i + 32]
            cols = list(zip(*ch))
            call("tacorl_copy_cols_batch", len(ch), (C.c_void_p * len(ch))(*cols[0]), int_array(cols[1]),
                 (C.c_void_p * len(ch))(*cols[2]), int_array(cols[3]), int_array(cols[4]), int_array(cols[5]),
                 int_array(cols[6]), int_array(cols[7]), stream())
        return False


def copy_cols(src, src_off, ld_src, dst, dst_off, ld_dst, rows, cols, src_row_mod=0, accumulate=False):
    if _copy_batch is not None:
        _copy_batch.append((src.data_ptr() + 4 * src_off, ld_src, dst.data_ptr() + 4 * dst_off, ld_dst, rows, cols,
                            src_row_mod, int(accumulate)))
        return
    call("tacorl_copy_cols", _at(src, src_off), ld_src, _at(dst, dst_off), ld_dst, rows, cols, src_row_mod,
         int(accumulate), stream())


def reduce_rows_mod(src, src_off, ld_in, dst, dst_off, ld_out, B, cols, reps):
    call("tacorl_reduce_rows_mod", _at(src, src_off), ld_in, _at(dst, dst_off), ld_out, B, cols, reps, stream())


def uniform_actions(u01, dst, dst_off, ld_dst, rows, A, discrete_gripper):
    call("tacorl_uniform_actions", ptr(u01), _at(dst, dst_off), ld_dst, rows, A, int(discrete_gripper), stream())


def tanh_normal_sample(head, ld_head, eps, gumbel_u, hard, act_out, act_off, ld_act, logp, grip_idx, n, M, Ac):
    call("tacorl_tanh_normal_sample", ptr(head), ld_head, ptr(eps), ptr(gumbel_u), int(hard), _at(act_out, act_off),
         ld_act, ptr(logp), ptr(grip_idx), n, M, Ac, stream())


def adam_step_batch(items, mirrors=None):
    """items: list of (param, grad, m, v, lr, max_norm, step_counter, target_or_None, tau) - one
    norm (+ step counter) launch and one update launch for all of them.
    mirrors: optional list of (bf16 mirror of param or None, bf16 mirror of target or None), written by the update launch."""
    k = len(items)
    nb = L.lib().tacorl_adam_batch_ws_bytes(k)
    ws = workspace(nb, items[0][0].device, "adam_batch")
    mir = ptr_array([m_[0] for m_ in mirrors]) if mirrors else None
    tmir = ptr_array([m_[1] for m_ in mirrors]) if mirrors else None
    call("tacorl_adam_step_batch_mirror", k, ptr_array([i[0] for i in items]), ptr_array([i[1] for i in items]),
         ptr_array([i[2] for i in items]), ptr_array([i[3] for i in items]), (C.c_long * k)(*[i[0].numel() for i in items]),
         (C.c_float * k)(*[float(i[4]) for i in items]), (C.c_float * k)(*[float(i[5]) for i in items]),
         ptr_array([i[6] for i in items]), ptr_array([i[7] for i in items]), (C.c_float * k)(*[float(i[8]) for i in items]),
         mir, tmir, ptr(ws), ws.numel(), stream())
    touched(*[i[0] for i in items], *[i[7] for i in items if i[7] is not None])


def touched(*tensors):
    """The library's kernels write parameter blocks through raw pointers, which torch's version counters do not see;
    bump them by hand (host-side, no launch) so that caches keyed on `tensor._version` - the frozen-network weight
    preparation - also notice optimiser steps.  Views share their base's counter."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


def adam_step(param, grad, m, v, lr, max_norm, step_counter, target=None, tau=0.0):
    n = param.numel()
    nb = L.lib().tacorl_adam_ws_bytes(n)
    ws = workspace(nb, param.device, "adam")
    call("tacorl_adam_step", ptr(param), ptr(grad), ptr(m), ptr(v), n, float(lr), float(max_norm), ptr(step_counter),
         ptr(target), float(tau), ptr(ws), ws.numel(), stream())
    touched(param, *([target] if target is not None else []))
