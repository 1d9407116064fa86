"""ctypes binding of libomni_cdna4.so — the C-ABI declared in include/omni_cdna4.h.

The product path has NO CPU / PyTorch fallback: if the shared library is missing or a call returns a
non-zero status, an exception is raised (the worker turns it into `DiffusionOutput.error`, mirroring
the reference's error path at vllm_omni/diffusion/worker/gpu_worker.py:266-274).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libomni_cdna4.so")   # fixed: dev sweeps assign this attribute (tools/devlib.py)
ABI_VERSION = 13

c_bf16_p = C.c_void_p  # device pointer to uint16_t bf16 bits
c_i32_p = C.c_void_p
c_f32_p = C.c_void_p


class OmniNativeError(RuntimeError):
    pass


class GemmGroup(C.Structure):
    _fields_ = [
        ("A", c_bf16_p), ("lda", C.c_int64), ("a_row_map", c_i32_p), ("M", C.c_int32),
        ("W", c_bf16_p), ("bias", c_bf16_p),
        ("out", c_bf16_p), ("out1", c_bf16_p), ("out2", c_bf16_p), ("ldo", C.c_int64),
        ("out_row_map", c_i32_p),
        ("res", c_bf16_p), ("ldres", C.c_int64),
        ("gate", c_bf16_p), ("gate_item_stride", C.c_int64),
        ("row_item_map", c_i32_p), ("rows_per_item", C.c_int32),
        ("qk_norm_q_w", c_bf16_p), ("qk_norm_k_w", c_bf16_p), ("qk_rope_cos", c_bf16_p), ("qk_rope_sin", c_bf16_p),
        ("qk_row_pos", c_i32_p), ("qk_eps", C.c_float),
        ("a_k32_rows", C.c_int32), ("out_k32_rows", C.c_int32), ("qk_q_scale", C.c_float),      # qk_q_scale: ABI v5
        ("tile_skip", c_i32_p),
        ("a_scale", C.c_void_p), ("w_scale", C.c_void_p),                                        # ABI v7 (fp8)
    ]


class GemmParams(C.Structure):
    _fields_ = [
        ("ngroups", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("epilogue", C.c_int32),
        ("split_n", C.c_int32), ("w_k32_blocked", C.c_int32), ("g", GemmGroup * 2),
        ("splitk_ws", C.c_void_p), ("splitk_ws_floats", C.c_int64),          # ABI v4
        ("kernel_hint", C.c_int32), ("fp8", C.c_int32),                      # ABI v6: kernel_hint; v7: fp8 operands
    ]


class AdalnStream(C.Structure):
    """omni_adaln_stream (ABI v12): one row group of omni_adaln_modulate_pair."""
    _fields_ = [
        ("x", c_bf16_p), ("y", c_bf16_p), ("rows", C.c_int32), ("scale", c_bf16_p), ("shift", c_bf16_p),
        ("row_item_map", c_i32_p), ("rows_per_item", C.c_int32), ("y_k32_rows", C.c_int32),
        ("y8", C.c_void_p), ("y8_rows", C.c_int32), ("y8_scale", C.c_void_p),
    ]


class FinishAdalnStream(C.Structure):
    """omni_finish_adaln_stream (ABI v13): one row group of omni_splitk_finish_adaln_pair."""
    _fields_ = [
        ("rows", C.c_int32), ("ws_row0", C.c_int32), ("bias", c_bf16_p), ("hidden", c_bf16_p), ("gate", c_bf16_p),
        ("scale", c_bf16_p), ("shift", c_bf16_p), ("row_item_map", c_i32_p), ("rows_per_item", C.c_int32),
        ("y", c_bf16_p), ("y_k32_rows", C.c_int32),
    ]


class AttnParams(C.Structure):
    """omni_attn_params (ABI v11): the general attention of the SDPA plug-in point — cross-attention, masks, causal, dh 64 / 128."""
    _fields_ = [
        ("q", c_bf16_p), ("k", c_bf16_p), ("v", c_bf16_p), ("out", c_bf16_p),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
        ("cu_seqlens_q", c_i32_p), ("cu_seqlens_k", c_i32_p),
        ("B", C.c_int32), ("H", C.c_int32), ("H_kv", C.c_int32), ("head_dim", C.c_int32),
        ("max_seqlen_q", C.c_int32), ("max_seqlen_k", C.c_int32),
        ("softmax_scale", C.c_float), ("causal", C.c_int32),
        ("mask", C.c_void_p), ("mask_type", C.c_int32),
        ("mask_stride_b", C.c_int64), ("mask_stride_h", C.c_int64), ("mask_stride_q", C.c_int64), ("mask_stride_k", C.c_int64),
    ]


class ConvParams(C.Structure):
    _fields_ = [
        ("x", c_bf16_p), ("w", c_bf16_p), ("bias", c_bf16_p), ("gamma", c_bf16_p), ("res", c_bf16_p), ("y", c_bf16_p),
        ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("ksize", C.c_int32), ("upsample2x", C.c_int32), ("silu", C.c_int32),
        ("clamp_lo", C.c_float), ("clamp_hi", C.c_float), ("downsample2x", C.c_int32),
        ("x_padded", C.c_int32), ("y_padded", C.c_int32),
        ("norm_gamma", c_bf16_p), ("y_norm", c_bf16_p), ("norm_silu", C.c_int32),                 # ABI v10
    ]


_LAYER_FIELDS = [
    "img_mod_w", "img_mod_b", "txt_mod_w", "txt_mod_b",
    "to_qkv_w", "to_qkv_b", "add_qkv_w", "add_qkv_b",
    "norm_q_w", "norm_k_w", "norm_added_q_w", "norm_added_k_w",
    "to_out_w", "to_out_b", "to_add_out_w", "to_add_out_b",
    "img_mlp_w1", "img_mlp_b1", "img_mlp_w2", "img_mlp_b2",
    "txt_mlp_w1", "txt_mlp_b1", "txt_mlp_w2", "txt_mlp_b2",
]


class DitLayerWeights(C.Structure):
    _fields_ = [(n, c_bf16_p) for n in _LAYER_FIELDS]


_FP8_FIELDS = ["to_qkv", "add_qkv", "to_out", "to_add_out", "img_mlp_w1", "img_mlp_w2", "txt_mlp_w1", "txt_mlp_w2"]


class DitFp8Layer(C.Structure):
    """omni_dit_fp8_layer: e4m3 copies (K64-blocked) of a layer's eight GEMM weights, then their per-channel fp32 scales."""
    _fields_ = [(f + ("_w8" if "mlp" not in f else "_8"), C.c_void_p) for f in _FP8_FIELDS] + [(f + "_s", C.c_void_p) for f in _FP8_FIELDS]


class DitWeights(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("head_dim", C.c_int32), ("joint_dim", C.c_int32),
        ("in_channels", C.c_int32), ("out_channels_packed", C.c_int32), ("gemm_w_k32_blocked", C.c_int32),
        ("t_lin1_w", c_bf16_p), ("t_lin1_b", c_bf16_p), ("t_lin2_w", c_bf16_p), ("t_lin2_b", c_bf16_p),
        ("txt_norm_w", c_bf16_p), ("img_in_w", c_bf16_p), ("img_in_b", c_bf16_p), ("txt_in_w", c_bf16_p),
        ("txt_in_b", c_bf16_p),
        ("norm_out_w", c_bf16_p), ("norm_out_b", c_bf16_p), ("proj_out_w", c_bf16_p), ("proj_out_b", c_bf16_p),
        ("layers", C.POINTER(DitLayerWeights)),
        ("fp8_layers", C.POINTER(DitFp8Layer)),                                                    # ABI v7, nullable
    ]


class TeaCache(C.Structure):
    """omni_teacache: device-side TeaCache state of one step-batch (include/omni_cdna4.h)."""
    _fields_ = [
        ("rel_l1_thresh", C.c_float), ("coeff", C.c_float * 5),
        ("prev_mod", c_bf16_p), ("prev_res", c_bf16_p), ("acc_dist", c_f32_p), ("cnt", c_i32_p), ("skip", c_i32_p),
        ("skip_total", c_i32_p), ("scratch", c_f32_p), ("tile_skip_img", c_i32_p), ("tile_skip_txt", c_i32_p),
        ("txt_cu", c_i32_p),
    ]


class DitBatch(C.Structure):
    _fields_ = [
        ("n_items", C.c_int32), ("n_img_rows", C.c_int32), ("n_txt_rows", C.c_int32), ("n_joint_rows", C.c_int32),
        ("n_temb", C.c_int32), ("max_seqlen", C.c_int32),
        ("latents", c_bf16_p), ("prompt_embeds", c_bf16_p), ("timestep", c_f32_p),
        ("cu_seqlens", c_i32_p), ("img_item", c_i32_p), ("txt_item", c_i32_p),
        ("img_joint_row", c_i32_p), ("txt_joint_row", c_i32_p), ("joint_pos", c_i32_p),
        ("txt_pos_end", C.c_int32),
        ("rope_cos", c_bf16_p), ("rope_sin", c_bf16_p),
        ("noise_pred", c_bf16_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("teacache", C.POINTER(TeaCache)),
        ("temb_add", c_bf16_p),                                                                    # ABI v9, nullable
        ("mod_table", c_bf16_p),                                                                   # ABI v9, nullable
    ]


# name -> (restype, argtypes); must list every symbol include/omni_cdna4.h declares
PROTOTYPES = {
    "omni_abi_version": (C.c_int, []),
    "omni_build_arch": (C.c_char_p, []),
    "omni_status_string": (C.c_char_p, [C.c_int]),
    "omni_gemm_bf16": (C.c_int, [C.POINTER(GemmParams), C.c_void_p]),
    "omni_quantize_fp8_rows": (C.c_int, [c_bf16_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p]),
    "omni_adaln_modulate": (C.c_int, [c_bf16_p, C.c_int64, c_bf16_p, C.c_int64, C.c_int32, C.c_int32, c_bf16_p,
                                      c_bf16_p, C.c_int64, c_i32_p, C.c_int32, C.c_float, C.c_void_p]),
    "omni_adaln_modulate_ex": (C.c_int, [c_bf16_p, C.c_int64, c_bf16_p, C.c_int64, C.c_int32, C.c_int32, c_bf16_p,
                                         c_bf16_p, C.c_int64, c_i32_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "omni_adaln_modulate_fp8": (C.c_int, [c_bf16_p, C.c_int64, C.c_int32, C.c_int32, c_bf16_p, c_bf16_p, C.c_int64, c_i32_p,
                                          C.c_int32, C.c_float, c_bf16_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_void_p]),
    "omni_adaln_modulate_pair": (C.c_int, [C.POINTER(AdalnStream), C.POINTER(AdalnStream), C.c_int32, C.c_int64, C.c_float,
                                           C.c_void_p]),                                                        # ABI v12
    "omni_gemm_splitk_factor": (C.c_int, [C.POINTER(GemmParams)]),                                              # ABI v13
    "omni_splitk_finish_adaln_pair": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(FinishAdalnStream),
                                                C.POINTER(FinishAdalnStream), C.c_int32, C.c_int64, C.c_float, C.c_void_p]),  # ABI v13
    "omni_rmsnorm": (C.c_int, [c_bf16_p, C.c_int64, c_bf16_p, C.c_int64, C.c_int32, C.c_int32, c_bf16_p, C.c_float,
                               C.c_void_p]),
    "omni_qk_norm_rope": (C.c_int, [c_bf16_p, C.c_int64, C.c_int32, C.c_int32, c_bf16_p, c_bf16_p, c_bf16_p, c_bf16_p,
                                    c_i32_p, C.c_int32, C.c_float, C.c_void_p]),
    "omni_rope_interleaved": (C.c_int, [c_bf16_p, c_bf16_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_bf16_p,
                                        c_bf16_p, C.c_void_p]),
    "omni_flash_attn_fwd": (C.c_int, [c_bf16_p, c_bf16_p, c_bf16_p, c_bf16_p, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int64, c_i32_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                      C.c_void_p]),
    "omni_flash_attn_fwd_ex": (C.c_int, [c_bf16_p, c_bf16_p, c_bf16_p, c_bf16_p, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_int64, c_i32_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                         C.c_int32, C.c_void_p]),
    "omni_flash_attn_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),                         # ABI v11
    "omni_flash_attn_fwd_ws": (C.c_int, [c_bf16_p, c_bf16_p, c_bf16_p, c_bf16_p, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_int64, c_i32_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                         C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),         # ABI v11
    "omni_flash_attn_general": (C.c_int, [C.POINTER(AttnParams), C.c_void_p]),                      # ABI v11
    "omni_linear_smallbatch": (C.c_int, [c_bf16_p, C.c_int64, C.c_int32, c_bf16_p, c_bf16_p, C.c_int64, C.c_int32,
                                         c_bf16_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "omni_timestep_sinusoid": (C.c_int, [c_f32_p, C.c_int32, C.c_int32, C.c_float, c_bf16_p, C.c_void_p]),
    "omni_cfg_euler_step": (C.c_int, [c_bf16_p, c_bf16_p, c_bf16_p, C.c_int32, C.c_int32, C.c_float, c_f32_p,
                                      C.c_int32, C.c_void_p]),
    "omni_cfg_euler_step_ex": (C.c_int, [c_bf16_p, c_bf16_p, c_bf16_p, C.c_int32, C.c_int32, C.c_float, c_f32_p,
                                         C.c_int32, C.c_int32, C.c_void_p]),                      # ABI v9
    "omni_vae_conv2d": (C.c_int, [C.POINTER(ConvParams), C.c_void_p]),
    "omni_vae_conv2d_fuses_norm": (C.c_int, [C.POINTER(ConvParams)]),                              # ABI v10
    "omni_vae_upsample2x_bordered": (C.c_int, [c_bf16_p, c_bf16_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "omni_vae_rmsnorm_silu": (C.c_int, [c_bf16_p, c_bf16_p, C.c_int64, C.c_int32, c_bf16_p, C.c_int32, C.c_void_p]),
    "omni_softmax_rows": (C.c_int, [c_bf16_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "omni_vae_attention": (C.c_int, [c_bf16_p, c_bf16_p, c_bf16_p, c_bf16_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                     C.c_int64, C.c_int64, C.c_float, C.c_void_p]),                 # ABI v10
    "omni_dit_workspace_bytes": (C.c_size_t, [C.POINTER(DitWeights), C.c_int32, C.c_int32, C.c_int32]),
    "omni_dit_modulation_table_workspace_bytes": (C.c_size_t, [C.POINTER(DitWeights), C.c_int32]),
    "omni_dit_modulation_table": (C.c_int, [C.POINTER(DitWeights), c_bf16_p, C.c_int32, c_bf16_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    "omni_dit_forward": (C.c_int, [C.POINTER(DitWeights), C.POINTER(DitBatch), C.c_void_p]),
    "omni_dit_block": (C.c_int, [C.POINTER(DitWeights), C.c_int32, C.POINTER(DitBatch), c_bf16_p, c_bf16_p, c_bf16_p,
                                 C.c_void_p]),
    "omni_dit_block_qkv": (C.c_int, [C.POINTER(DitWeights), C.c_int32, C.POINTER(DitBatch), c_bf16_p, c_bf16_p, c_bf16_p,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "omni_dit_block_post": (C.c_int, [C.POINTER(DitWeights), C.c_int32, C.POINTER(DitBatch), c_bf16_p, c_bf16_p, c_bf16_p,
                                      c_bf16_p, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Load (once) and return the native library; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise OmniNativeError(
                f"{LIB_PATH} is missing: build it with `python vllm_omni_amd/csrc/build.py` "
                "(or __graft_entry__.build()).  There is no CPU fallback for the DiT hot path.")
        dll = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(dll, name)
            except AttributeError as e:  # pragma: no cover
                raise OmniNativeError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        got = dll.omni_abi_version()
        if got != ABI_VERSION:
            raise OmniNativeError(f"ABI mismatch: library {got}, binding {ABI_VERSION}")
        _lib = dll
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().omni_status_string(status).decode()
        raise OmniNativeError(f"{what} failed: status {status} ({msg})")
