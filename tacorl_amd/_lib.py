"""ctypes binding of libtacorl_hip.so (C ABI: include/tacorl_hip.h).

There is no CPU or eager-torch fallback: if the library is missing or a call fails,
this raises.  Everything in tacorl_amd that computes goes through `call()`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (TACORL_HIP_LIB: another build of the same library - same-box A/B of a kernel change, scratch/mklib_file.sh)
LIB_PATH = os.environ.get("TACORL_HIP_LIB") or os.path.join(HERE, "lib", "libtacorl_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2
MAXP = 16

LOG_SLOTS = [  # enum TACORL_LG_* -> reference self.log key
    "alpha_loss", "alpha", "actor_loss", "bellman_q1_loss", "bellman_q2_loss", "conservative_q1_loss",
    "conservative_q2_loss", "q1_loss", "q2_loss", "alpha_prime", "alpha_prime_loss", "q1_data", "q1_random",
    "q1_policy", "q2_data", "q2_random", "q2_policy", "action_loss",
]

_p, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
_SIGS = {
    "tacorl_hip_version": (_i, []),
    "tacorl_hip_init": (_i, [_i]),
    "tacorl_hip_last_error": (C.c_char_p, []),
    "tacorl_time_mark": (_i, [_p, _i, _p]),
    "tacorl_time_spin": (_i, [_p, _i, _l, _p]),
    "tacorl_linear_fwd": (_i, [_i, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_linear_add_fwd_ws_bytes": (_sz, [_i, _p, _i, _i]),
    "tacorl_linear_add_fwd": (_i, [_i, _p, _i, _p, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "tacorl_rnn_linear_supported": (_i, [_i, _i, _i]),
    "tacorl_rnn_linear_fwd": (_i, [_p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_rnn_linear_fwd_batch": (_i, [_i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _p]),
    "tacorl_rnn_linear_fwd_batch_twin": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "tacorl_rnn_linear_fwd_batch_ext": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "tacorl_rnn_linear_bwd_step": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_rnn_linear_bwd_batch": (_i, [_i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_rnn_wgrad_supported": (_i, [_i, _i, _i]),
    "tacorl_rnn_wgrad": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p, _i, _p]),
    "tacorl_rnn_wgrad_batch": (_i, [_i, _p, _i, _p, _i, _p, _i, _i, _p, _p, _i, _p]),
    "tacorl_rnn_wgrad_slabs_ws_bytes": (_sz, [_i, _i, _i]),
    "tacorl_rnn_wgrad_slabs": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _sz, _p]),
    "tacorl_transpose_to_bf16": (_i, [_p, _p, _i, _i, _p]),
    "tacorl_transpose_to_bf16_batch": (_i, [_i, _p, _p, _p, _p, _p]),
    "tacorl_transpose_pad_to_bf16": (_i, [_p, _p, _i, _i, _i, _p]),
    "tacorl_pad_to_bf16": (_i, [_p, _i, _p, _i, _l, _i, _p]),
    "tacorl_pr_encoder_fused_supported": (_i, [_i, _i, _i, _i, _i]),
    "tacorl_pr_encoder_fused_train_supported": (_i, [_i, _i, _i, _i, _i]),
    "tacorl_pr_encoder_fused": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "tacorl_pr_encoder_fused_train": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "tacorl_pr_encoder_bwd_fused": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "tacorl_pr_encoder_fused_train_sample": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _f, _p]),
    "tacorl_pr_encoder_fused_sample": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _f, _p]),
    "tacorl_pr_head_compose": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_to_bf16_batch": (_i, [_i, _p, _p, _p, _p]),
    "tacorl_mlp_bwd_fused_supported": (_i, [_i, _i, _p, _i, _i]),
    "tacorl_mlp_bwd_fused_ws_bytes": (_sz, [_i, _p, _i, _p]),
    "tacorl_mlp_bwd_fused_dgrad": (_i, [_i, _p, _p, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _sz, _p]),
    "tacorl_mlp_bwd_fused_pack": (_i, [_i, _p, _p, _i, _p, _p, _sz, _p]),
    "tacorl_prep_batch_begin": (_i, []),
    "tacorl_reduce_batch_begin": (_i, []),
    "tacorl_reduce_batch_end": (_i, [_p]),
    "tacorl_prep_batch_end": (_i, [_p]),
    "tacorl_mlp_bwd_fused_wgrad": (_i, [_i, _p, _i, _p, _p, _i, _p, _p, _i, _p, _p, _i, _i, _p, _sz, _p]),
    "tacorl_mlp_fwd_fused_supported": (_i, [_i, _i, _p, _i]),
    "tacorl_mlp_lean_supported": (_i, [_i, _i, _p, _i, _i, _i]),
    "tacorl_mlp_fwd_fused": (_i, [_i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _i, _p]),
    "tacorl_mlp_fwd_fused_gather_supported": (_i, [_i, _p, _i, _p, _p, _i, _i]),
    "tacorl_mlp_fwd_fused_gather": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _i, _p]),
    "tacorl_add_rows_bcast": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_attention_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "tacorl_attention_dropout_fwd": (_i, [_p, _p, _p, _f, _i, _i, _i, _i, _p]),
    "tacorl_attention_dropout_bwd": (_i, [_p, _p, _p, _p, _f, _i, _i, _i, _i, _p]),
    "tacorl_dropout_mul": (_i, [_p, _p, _f, _l, _p]),
    "tacorl_add_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "tacorl_mean_over_t": (_i, [_p, _p, _i, _i, _i, _p]),
    "tacorl_pr_sample": (_i, [_p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "tacorl_build_ad_input": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "tacorl_build_ad_input_bf16": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "tacorl_ad_input_proj": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "tacorl_logistic_mixture_ws_bytes": (_sz, [_i, _i, _i]),
    "tacorl_logistic_mixture_loss": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _p, _sz, _p]),
    "tacorl_logistic_mixture_finish": (_i, [_p, _sz, _i, _i, _i, _p, _p]),
    "tacorl_logistic_mixture_sample": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_linear_dgrad": (_i, [_i, _p, _i, _p, _p, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i, _p]),
    "tacorl_linear_dgrad_ws_bytes": (_sz, [_i, _p, _i, _i]),
    "tacorl_linear_dgrad_splitk": (_i, [_i, _p, _i, _p, _p, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i, _p, _sz, _p]),
    "tacorl_linear_wgrad_ws_bytes": (_sz, [_i, _p, _i, _i]),
    "tacorl_linear_wgrad": (_i, [_i, _p, _i, _p, _i, _p, _i, _i, _p, _p, _i, _i, _p, _sz, _p]),
    "tacorl_relu_mask_mul": (_i, [_p, _p, _p, _p, _l, _p]),
    "tacorl_ad_input_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tacorl_plmp_demb_finish": (_i, [_p, _p, _i, _i, _p, _i, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_bcast_over_t": (_i, [_p, _p, _i, _i, _i, _f, _i, _p]),
    "tacorl_attention_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_add_layernorm_bwd_ws_bytes": (_sz, [_i, _i]),
    "tacorl_add_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "tacorl_gauss_kl_balanced": (_i, [_p, _p, _p, _p, _i, _i, _f, _f, _f, _i, _f, _p, _p]),
    "tacorl_pr_sample_bwd": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "tacorl_conv2d_relu_fwd": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tacorl_encoder_param_layout": (_l, [_p]),
    "tacorl_encoder_act_layout": (_l, [_i, _i, _i, _p]),
    "tacorl_encoder_fwd": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_encoder_bwd_ws_bytes": (_sz, [_i, _p, _i, _i]),
    "tacorl_encoder_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "tacorl_encoder_fused_wpk_bytes": (_l, []),
    "tacorl_encoder_fused_supported": (_i, [_i, _i]),
    "tacorl_encoder_fused_act_format": (_i, [_i, _i]),
    "tacorl_encoder_pack_weights": (_i, [_i, _p, _p, _p]),
    "tacorl_encoder_fwd_fused": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "tacorl_encoder_fwd_fused_wg": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_encoder_bwd_fused_ws_bytes": (_sz, [_i, _p, _i, _i]),
    "tacorl_encoder_bwd_fused": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "tacorl_encoder_bwd_fused_pack": (_i, [_i, _p, _p, _i, _i, _p, _sz, _p]),
    "tacorl_encoder_bwd_fused_head": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "tacorl_encoder_bwd_fused_fc_wgrad": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "tacorl_encoder_bwd_fused_conv": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "tacorl_encoder_bwd_fused_conv_parts": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "tacorl_mlp_param_layout": (_l, [_i, _p, _p, _p]),
    "tacorl_mlp_act_layout": (_l, [_i, _i, _p, _p, _p, _p]),
    "tacorl_mlp_fwd": (_i, [_i, _p, _i, _p, _p, _p, _i, _p, _p, _i, _p]),
    "tacorl_mlp_bwd_ws_bytes": (_sz, [_i, _p, _i, _p]),
    "tacorl_mlp_bwd": (_i, [_i, _p, _i, _p, _p, _p, _i, _p, _p, _i, _p, _i, _p, _p, _i, _i, _p, _sz, _p]),
    "tacorl_pack_images": (_i, [_p, _l, _i, _p, _i, _i, _i, _i, _i, _p]),
    "tacorl_pack_images_batch": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_pack_images_window_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_pack_images_u8_batch": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_pack_images_u8_gather_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "tacorl_pack_images_u8_aug_gather_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_pack_images_u8_resize_aug_gather_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "tacorl_gather_frames_u8": (_i, [_p, _l, _p, _p, _l, _p]),
    "tacorl_pack_images_u8_aug_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "tacorl_stage_transition": (_i, [_p, _i, _p, _p, _i, _p, _p, _l, _p]),
    "tacorl_copy_cols": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "tacorl_copy_cols_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "tacorl_reduce_rows_mod": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "tacorl_reduce_rows_mod_batch": (_i, [_i, _p, _i, _p, _i, _i, _i, _i, _p]),
    "tacorl_uniform_actions": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "tacorl_tanh_normal_sample": (_i, [_p, _i, _p, _p, _i, _p, _i, _p, _p, _i, _i, _i, _p]),
    "tacorl_tanh_normal_sample_batch": (_i, [_i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _p]),
    "tacorl_alpha_loss": (_i, [_p, _i, _p, _f, _f, _p, _p, _p]),
    "tacorl_alpha_loss_step": (_i, [_p, _i, _p, _f, _p, _p, _p, _p, _f, _p, _p]),
    "tacorl_actor_qmin": (_i, [_p, _p, _p, _i, _p, _p, _p, _f, _p, _p]),
    "tacorl_actor_head_bwd": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _f, _p, _i, _i, _i, _p, _p]),
    "tacorl_cql_ws_bytes": (_sz, [_i]),
    "tacorl_cql_loss": (_i, [_p] * 13 + [_i, _i, _i, _f, _f, _f, _f, _f, _i, _f, _p, _p, _p, _sz, _p]),
    "tacorl_adam_ws_bytes": (_sz, [_l]),
    "tacorl_adam_step": (_i, [_p, _p, _p, _p, _l, _f, _f, _p, _p, _f, _p, _sz, _p]),
    "tacorl_adam_batch_ws_bytes": (_sz, [_i]),
    "tacorl_adam_step_batch": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "tacorl_adam_step_batch_mirror": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
}

_lib = None


class TacorlHipError(RuntimeError):
    pass


def lib():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # torch ships its own libamdhip64; import it first so this library binds to the SAME HIP
        # runtime instance (streams and device pointers are shared with torch).
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise TacorlHipError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -m tacorl_amd.build`.")
        L = C.CDLL(LIB_PATH)
        override = bool(os.environ.get("TACORL_HIP_LIB"))
        missing = []
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name, None)
            if fn is None:
                missing.append(name)  # (the product library: __graft_entry__.build / tests/test_abi_cpu.py report it)
                continue
            fn.restype, fn.argtypes = res, args
        if override:
            # another build stands in for the product library in this process (scratch A/B builds): say so, and refuse one
            # whose entry points are not this binding's - a stale scratch .so with an older argument list would not fail
            # here but corrupt memory at its first call
            import sys

            if missing:
                raise TacorlHipError(f"TACORL_HIP_LIB={LIB_PATH} lacks {len(missing)} entry point(s) of this binding "
                                     f"({', '.join(missing[:5])}...): rebuild it from this tree")
            print(f"[tacorl_amd] TACORL_HIP_LIB override active: {LIB_PATH}", file=sys.stderr, flush=True)
        _lib = L
    return _lib


def declared_symbols():
    return list(_SIGS)


def call(name, *args):
    fn = getattr(lib(), name)
    rc = fn(*args)
    if fn.restype is _i and rc != 0:
        raise TacorlHipError(f"{name} failed ({rc}): {lib().tacorl_hip_last_error().decode()}")
    return rc


# ----------------------------------------------------------------- small marshalling helpers
def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """void*[n] from tensors / raw ints / None."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t is None:
            arr[i] = None
        elif isinstance(t, int):
            arr[i] = t
        elif isinstance(t, C.c_void_p):
            arr[i] = t.value
        else:
            arr[i] = t.data_ptr()
    return arr


def int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
