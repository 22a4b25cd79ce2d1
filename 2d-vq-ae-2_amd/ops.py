"""Functional wrappers over the C ABI for torch tensors living in HBM (plumbing only: pointers,
shapes, the current HIP stream).  Activations are NHWC fp32 unless a name says otherwise."""
import ctypes

import torch

from . import _lib as L

_IDX_DTYPES = {torch.int64: L.IDX_I64, torch.uint8: L.IDX_U8, torch.int32: L.IDX_I32}
if hasattr(torch, "uint16"):
    _IDX_DTYPES[torch.uint16] = L.IDX_U16


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.VqaeHipError("libvqae_hip ops need tensors in HBM (device='cuda'); there is no CPU fallback")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def idx_code(dtype):
    try:
        return _IDX_DTYPES[dtype]
    except KeyError:
        raise AssertionError(f"unsupported index dtype {dtype}")


def nchw_to_nhwc(x):
    _need_gpu(x)
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device)
    L.check(L.lib().vqae_nchw_to_nhwc_f32(_p(x), B, C, H, W, _p(y), _stream()))
    return y


def nhwc_to_nchw(x):
    _need_gpu(x)
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    L.check(L.lib().vqae_nhwc_to_nchw_f32(_p(x), B, C, H, W, _p(y), _stream()))
    return y


def pack_conv_weight(w, dtype=None):
    """[cout][cin][k][k] (PyTorch) -> the MFMA kernel's packed layout; `dtype` ('bf16'/'f16') rounds the
    weights to that type (autocast casts conv weights)."""
    _need_gpu(w)
    w = w.contiguous().float()
    cout, cin, k, _ = w.shape
    n = L.lib().vqae_conv_packed_floats(cout, cin, k)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    L.check(L.lib().vqae_conv_pack_weight_f32(_p(w), cout, cin, k, _p(out), _stream()))
    L.check(L.lib().vqae_round_inplace_f32(_p(out), n, L.dtype_code(dtype), _stream()))
    return out


def conv2d(x, w_packed, cout, ksize, stride=1, pad=0, pad_mode=L.PAD_NONE, pre=None, act=None, scale_bias=None,
           bias_s=None, bias_vec=None, residual=None, out=None, dtype=None, gate=None):
    """NHWC fp32 conv through vqae_conv2d_f32.  pre = (a,) or (a, b); act = (a, b) [ELU form] or 'silu';
    scale_bias = (s, b); gate [B, cin]: per-image channel gate applied while x is loaded (vqae_conv2d_gated_f32)."""
    _need_gpu(x, w_packed)
    x = x.contiguous()
    B, H, W, cin = x.shape
    a = L.ConvArgs()
    a.batch, a.in_h, a.in_w, a.cin, a.cout = B, H, W, cin, cout
    a.ksize, a.stride, a.pad, a.pad_mode = ksize, stride, pad, pad_mode
    a.dtype = L.dtype_code(dtype)
    if pre is not None:
        if len(pre) == 1:
            a.pre_mode, a.pre_a = L.PRE_BIAS, float(pre[0])
        else:
            a.pre_mode, a.pre_a, a.pre_b = L.PRE_BIAS_ELU_BIAS, float(pre[0]), float(pre[1])
    if isinstance(act, str):
        assert act == "silu", act
        a.has_act = L.ACT_SILU
    elif act is not None:
        a.has_act, a.act_a, a.act_b = L.ACT_ELU, float(act[0]), float(act[1])
    if scale_bias is not None:
        a.has_scale, a.scale, a.bias_s = 1, float(scale_bias[0]), float(scale_bias[1])
    elif bias_s is not None:
        a.has_bias_s, a.bias_s = 1, float(bias_s)
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    if gate is not None:
        assert pre is None and tuple(gate.shape) == (B, cin), (pre, gate.shape)
        _need_gpu(gate)
        a.pre_mode = L.PRE_CHANNEL_GATE
        L.check(L.lib().vqae_conv2d_gated_f32(ctypes.byref(a), _p(x), _p(gate.contiguous()), _p(w_packed), _p(bias_vec),
                                              _p(residual), _p(out), _stream()))
        return out
    L.check(L.lib().vqae_conv2d_f32(ctypes.byref(a), _p(x), _p(w_packed), _p(bias_vec), _p(residual), _p(out),
                                    _stream()))
    return out


def dwconv(x, w_taps, bias=None, mode=L.DW_SAME, silu=False, want_partial=False):
    """Depthwise conv on NHWC x [B,H,W,C]; w_taps [k*k, C].  mode: DW_SAME (3x3 circular), DW_DOWN (2x2 s2),
    DW_UP (ConvTranspose2d 2x2 s2).  -> y, or (y, strip partial sums) for se_gate."""
    _need_gpu(x, w_taps)
    x = x.contiguous()
    B, H, W, C = x.shape
    Ho, Wo = {L.DW_SAME: (H, W), L.DW_DOWN: (H // 2, W // 2), L.DW_UP: (2 * H, 2 * W)}[mode]
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.device)
    part = None
    if want_partial:
        part = torch.empty(int(L.lib().vqae_dw_partial_floats(B, Ho, Wo, C)), dtype=torch.float32, device=x.device)
    L.check(L.lib().vqae_dwconv_f32(_p(x), _p(w_taps.contiguous()), _p(bias), B, H, W, C, mode, int(silu), _p(y), _p(part),
                                    _stream()))
    return (y, part) if want_partial else y


def se_gate(partial, batch, out_h, out_w, fc0_w, fc0_b, fc2_w, fc2_b):
    """SELayer gate [B, C] from dwconv's strip sums (layers/misc.py:23-29)."""
    _need_gpu(partial, fc0_w, fc0_b, fc2_w, fc2_b)
    hidden, C = fc0_w.shape
    gate = torch.empty((batch, C), dtype=torch.float32, device=partial.device)
    L.check(L.lib().vqae_se_gate_f32(_p(partial), batch, out_h, out_w, C, _p(fc0_w.contiguous()), _p(fc0_b.contiguous()),
                                     hidden, _p(fc2_w.contiguous()), _p(fc2_b.contiguous()), _p(gate), _stream()))
    return gate


def pixel_shuffle2(x, c):
    """[B,H,W,4c] (a, b, c) -> [B,2H,2W,c]."""
    _need_gpu(x)
    x = x.contiguous()
    B, H, W, c4 = x.shape
    assert c4 == 4 * c
    y = torch.empty((B, 2 * H, 2 * W, c), dtype=torch.float32, device=x.device)
    L.check(L.lib().vqae_pixel_shuffle2_f32(_p(x), B, H, W, c, _p(y), _stream()))
    return y


def fixup_same_supported(c, h, w):
    return bool(L.lib().vqae_fixup_same_supported(c, h, w))


def fixup_same_block(x, w1p, w2p, w3p, scalars8, dtype=None):
    """Whole 'same' Fixup block in one launch (x NHWC [B,H,W,C]); scalars8 = (b1a,b1b,b2a,b2b,b3a,b3b,b4,scale)."""
    _need_gpu(x, w1p, w2p, w3p)
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    sc = (ctypes.c_float * 8)(*[float(v) for v in scalars8])
    L.check(L.lib().vqae_fixup_same_block_f32(_p(x), _p(y), _p(w1p), _p(w2p), _p(w3p), B, H, W, C, sc,
                                              L.dtype_code(dtype), _stream()))
    return y


def conv3x3_direct(x, w, bias, x_u8=None, mean255=None, inv_std255=None, dtype=None):
    """Stem conv (3x3, zero pad, bias); x NHWC fp32 or x_u8 NHWC uint8 (normalised on device)."""
    src = x if x_u8 is None else x_u8
    _need_gpu(src, w, bias)
    src = src.contiguous()
    B, H, W, cin = src.shape
    cout = w.shape[0]
    y = torch.empty((B, H, W, cout), dtype=torch.float32, device=src.device)
    m = (ctypes.c_float * 3)(*mean255) if mean255 is not None else None
    s = (ctypes.c_float * 3)(*inv_std255) if inv_std255 is not None else None
    L.check(L.lib().vqae_conv3x3_direct_f32(_p(x) if x_u8 is None else None, _p(x_u8) if x_u8 is not None else None,
                                            m, s, _p(w.contiguous()), _p(bias.contiguous()), B, H, W, cin, cout,
                                            _p(y), L.dtype_code(dtype), _stream()))
    return y


def bicubic_up2(x, pre_bias=0.0):
    _need_gpu(x)
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    L.check(L.lib().vqae_bicubic_up2_f32(_p(x), B, H, W, C, float(pre_bias), _p(y), _stream()))
    return y


def vq_forward(z_flat, embed, commitment_cost=1.0, idx_dtype=torch.int64, want_q=True, want_margin=False, p=4):
    """z_flat [N, D], embed [K, D] -> (q [N, D] | None, idx [N], loss 0-d, margin [N] | None).  p: the Minkowski exponent of the
    reference's cdist = the rank of the quantiser's input (vq.py:97,121-129): 4 for NCHW, 3 / 5 for [B, D, L] / [B, D, d, h, w]."""
    _need_gpu(z_flat, embed)
    z_flat = z_flat.contiguous()
    embed = embed.contiguous()
    N, D = z_flat.shape
    K = embed.shape[0]
    dev = z_flat.device
    idx = torch.empty(N, dtype=idx_dtype, device=dev)
    q = torch.empty_like(z_flat) if want_q else None
    loss = torch.zeros((), dtype=torch.float32, device=dev)
    margin = torch.empty(N, dtype=torch.float32, device=dev) if want_margin else None
    ws = torch.empty(L.lib().vqae_vq_workspace_bytes(N, K, D), dtype=torch.uint8, device=dev)
    L.check(L.lib().vqae_vq_forward_p_f32(_p(z_flat), _p(embed), N, K, D, int(p), float(commitment_cost), _p(idx),
                                          idx_code(idx_dtype), _p(q), _p(loss), _p(margin), _p(ws), _stream()))
    return q, idx, loss, margin


def vq_projected(x_flat, proj_in_w, proj_in_b, embed, proj_out_w, proj_out_b, commitment_cost=1.0, idx_dtype=torch.int64,
                 dtype=None, want_z=False, want_margin=False):
    """ProjectedEMAVectorQuantizer2d.forward on x_flat [N, C] (NHWC rows) in one launch (projection_dim 8):
    proj_in_w [8, C(,1,1)], proj_out_w [C, 8(,1,1)] as PyTorch stores them -> (out [N, C], idx [N], loss 0-d, z | None,
    margin | None).  With `dtype` ('bf16' / 'f16') the two convolutions round like torch.autocast."""
    _need_gpu(x_flat, proj_in_w, proj_in_b, embed, proj_out_w, proj_out_b)
    x_flat = x_flat.contiguous().float()
    N, C = x_flat.shape
    K, D = embed.shape
    dev = x_flat.device
    code = L.dtype_code(dtype)
    tdt = {L.DT_F32: None, L.DT_BF16: torch.bfloat16, L.DT_F16: torch.float16}[code]
    r = (lambda t: t.to(tdt).float()) if tdt is not None else (lambda t: t)
    wt_in = r(proj_in_w.reshape(D, C).float()).t().contiguous()
    w_out = r(proj_out_w.reshape(C, D).float()).contiguous()
    b_in, b_out = r(proj_in_b.float()).contiguous(), r(proj_out_b.float()).contiguous()
    idx = torch.empty(N, dtype=idx_dtype, device=dev)
    out = torch.empty_like(x_flat)
    z = torch.empty((N, D), dtype=torch.float32, device=dev) if want_z else None
    loss = torch.zeros((), dtype=torch.float32, device=dev)
    margin = torch.empty(N, dtype=torch.float32, device=dev) if want_margin else None
    ws = torch.empty(L.lib().vqae_vq_projected_workspace_bytes(N), dtype=torch.uint8, device=dev)
    L.check(L.lib().vqae_vq_projected_f32(_p(x_flat), _p(wt_in), _p(b_in), _p(embed.contiguous().float()), _p(w_out), _p(b_out),
                                          N, C, D, K, float(commitment_cost), code, _p(idx), idx_code(idx_dtype), _p(out),
                                          _p(z), _p(loss), _p(margin), _p(ws), _stream()))
    return out, idx, loss, z, margin


def embed_code(idx, embed):
    _need_gpu(idx, embed)
    idx = idx.contiguous()
    embed = embed.contiguous()
    K, D = embed.shape
    out = torch.empty(tuple(idx.shape) + (D,), dtype=torch.float32, device=idx.device)
    L.check(L.lib().vqae_embed_code_f32(_p(idx), idx_code(idx.dtype), _p(embed), idx.numel(), K, D, _p(out),
                                        _stream()))
    return out


def vq_code_stats(z_flat, idx, n_codes):
    _need_gpu(z_flat, idx)
    z_flat = z_flat.contiguous()
    idx = idx.contiguous()
    N, D = z_flat.shape
    counts = torch.empty(n_codes, dtype=torch.float32, device=z_flat.device)
    dw = torch.empty((n_codes, D), dtype=torch.float32, device=z_flat.device)
    L.check(L.lib().vqae_vq_code_stats_f32(_p(z_flat), _p(idx), idx_code(idx.dtype), N, n_codes, D, _p(counts),
                                           _p(dw), _stream()))
    return counts, dw


def vq_ema_update(embed, embed_avg, cluster_size, counts, dw, decay, laplace_alpha):
    _need_gpu(embed, embed_avg, cluster_size, counts, dw)
    K, D = embed.shape
    ws = torch.empty(16, dtype=torch.uint8, device=embed.device)
    L.check(L.lib().vqae_vq_ema_update_f32(_p(embed), _p(embed_avg), _p(cluster_size), _p(counts), _p(dw), K, D,
                                           float(decay), float(laplace_alpha), _p(ws), _stream()))


def label_maxpool(labels_u8, out=32):
    _need_gpu(labels_u8)
    labels_u8 = labels_u8.contiguous()
    B, H, W = labels_u8.shape
    y = torch.empty((B, out, out), dtype=torch.uint8, device=labels_u8.device)
    L.check(L.lib().vqae_label_maxpool_u8(_p(labels_u8), B, H, W, out, _p(y), _stream()))
    return y


def stitch_tiles(tiles, rc, grid):
    """tiles [n, th, tw] scattered into grid [gh, gw] at patch positions rc [n, 2] (int32), in place."""
    _need_gpu(tiles, rc, grid)
    tiles = tiles.contiguous()
    rc = rc.to(torch.int32).contiguous()
    n, th, tw = tiles.shape
    L.check(L.lib().vqae_stitch_tiles(_p(tiles), idx_code(tiles.dtype), _p(rc), n, th, tw, _p(grid),
                                      idx_code(grid.dtype), grid.shape[0], grid.shape[1], _stream()))
    return grid
