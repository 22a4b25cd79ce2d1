"""ctypes binding of libvqae_hip.so (include/vqae_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, this module
raises.  Error codes are mapped back to the exception types the reference raises for the same
conditions (SURVEY.md §8b "Error conventions").
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# VQAE_HIP_LIB: developer override to time an alternative build of the same library (kernel experiments)
LIB_PATH = os.environ.get("VQAE_HIP_LIB") or os.path.join(_HERE, "libvqae_hip.so")

# enums of vqae_hip.h
LAYOUT_NHWC, LAYOUT_NCHW = 0, 1
IDX_I64, IDX_U8, IDX_U16, IDX_I32 = 0, 1, 2, 3
PAD_NONE, PAD_CIRCULAR, PAD_ZEROS = 0, 1, 2
PRE_NONE, PRE_BIAS, PRE_BIAS_ELU_BIAS, PRE_CHANNEL_GATE = 0, 1, 2, 3
ACT_NONE, ACT_ELU, ACT_SILU = 0, 1, 2
DW_SAME, DW_DOWN, DW_UP = 0, 1, 2
BLOCK_FIXUP, BLOCK_MBCONV = 0, 1
DT_F32, DT_BF16, DT_F16 = 0, 1, 2
DTYPES = {"f32": DT_F32, "fp32": DT_F32, "float32": DT_F32, "bf16": DT_BF16, "bfloat16": DT_BF16,
          "f16": DT_F16, "fp16": DT_F16, "float16": DT_F16, "half": DT_F16}


def dtype_code(dt):
    """'bf16' / torch.bfloat16 / None ... -> VQAE_DT_* (autocast semantics, see include/vqae_hip.h)."""
    if dt is None:
        return DT_F32
    if isinstance(dt, int):
        return dt
    name = str(dt).replace("torch.", "")
    try:
        return DTYPES[name]
    except KeyError:
        raise AssertionError(f"unsupported compute dtype {dt}")


class VqaeHipError(RuntimeError):
    pass


class ConvArgs(Structure):
    _fields_ = [("batch", c_int), ("in_h", c_int), ("in_w", c_int), ("cin", c_int), ("cout", c_int),
                ("ksize", c_int), ("stride", c_int), ("pad", c_int), ("pad_mode", c_int),
                ("pre_mode", c_int), ("pre_a", c_float), ("pre_b", c_float),
                ("has_scale", c_int), ("has_bias_s", c_int), ("has_act", c_int),
                ("scale", c_float), ("bias_s", c_float), ("act_a", c_float), ("act_b", c_float), ("dtype", c_int)]


class Config(Structure):
    _fields_ = [("in_channels", c_int), ("stem", c_int), ("n_down", c_int), ("n_pre", c_int), ("n_post", c_int),
                ("n_enc", c_int), ("num_embeddings", c_int), ("projection_dim", c_int),
                ("commitment_cost", c_float), ("compute_dtype", c_int),
                ("block_kind", c_int), ("expand_ratio", c_int), ("se_divisor", c_int), ("bn_eps", c_float)]


class Tensor(Structure):
    _fields_ = [("name", c_char_p), ("data", c_void_p), ("numel", c_int64)]


# name -> (restype, argtypes); every symbol include/vqae_hip.h declares
SYMBOLS = {
    "vqae_last_error": (c_char_p, []),
    "vqae_build_info": (c_char_p, []),
    "vqae_vq_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "vqae_vq_forward_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqae_vq_forward_p_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqae_vq_projected_workspace_bytes": (c_size_t, [c_int64]),
    "vqae_vq_projected_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                      c_float, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqae_embed_code_f32": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "vqae_vq_code_stats_f32": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vqae_vq_ema_update_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                       c_void_p, c_void_p]),
    "vqae_conv_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "vqae_conv_pack_weight_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_conv2d_f32": (c_int, [POINTER(ConvArgs), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vqae_conv2d_gated_f32": (c_int, [POINTER(ConvArgs), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "vqae_dw_partial_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vqae_dwconv_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                c_void_p, c_void_p]),
    "vqae_se_gate_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "vqae_pixel_shuffle2_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_fixup_same_supported": (c_int, [c_int, c_int, c_int]),
    "vqae_fixup_same_block_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          POINTER(c_float), c_int, c_void_p]),
    "vqae_round_inplace_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    "vqae_conv3x3_direct_f32": (c_int, [c_void_p, c_void_p, POINTER(c_float), POINTER(c_float), c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "vqae_bicubic_up2_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "vqae_nchw_to_nhwc_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_nhwc_to_nchw_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_label_maxpool_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_stitch_tiles": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                  c_void_p]),
    "vqae_create": (c_int, [POINTER(Config), POINTER(Tensor), c_int, POINTER(c_void_p)]),
    "vqae_destroy": (None, [c_void_p]),
    "vqae_reserve": (c_int, [c_void_p, c_int, c_int, c_int]),
    "vqae_set_codebook": (c_int, [c_void_p, c_void_p]),
    "vqae_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                            c_void_p]),
    "vqae_encode_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                               c_void_p]),
    "vqae_encode_features": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_decode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_decode_indices": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vqae_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                             c_void_p]),
    "vqae_block_count": (c_int, [c_void_p, c_int]),
    "vqae_run_blocks": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, POINTER(c_int),
                                POINTER(c_int), c_void_p]),
    "vqae_flops_per_patch": (c_double, [c_void_p, c_int, c_int, c_int, c_int]),
    "vqae_prof_begin": (c_int, [c_int, c_int]),
    "vqae_prof_end": (c_int, [POINTER(c_double), POINTER(c_int), POINTER(c_double)]),
}

_lib = None


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VqaeHipError(f"{LIB_PATH} is missing: build it with 2d-vq-ae-2_amd/build.sh "
                               "(__graft_entry__.build()); there is no CPU fallback")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int):
    """Map a vqae_status to the reference's exception types."""
    if rc == 0:
        return
    msg = lib().vqae_last_error().decode(errors="replace")
    if rc == -1:
        raise AssertionError(msg)                 # reference: assert (vq.py:98, conv_block.py:148)
    if rc == -2:
        raise NotImplementedError(msg)            # reference: vq.py:100-104
    if rc == -5:
        raise KeyError(msg)                       # missing state-dict entry
    if rc == -4:
        raise MemoryError(msg)
    raise VqaeHipError(f"libvqae_hip error {rc}: {msg}")
