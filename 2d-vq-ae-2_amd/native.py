"""vqae_handle wrapper: the whole-model native runtime of libvqae_hip.so."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from . import ops
from .spec import VQAESpec

_BUFFER_ONLY = ("embed_avg", "cluster_size", "first_pass", "num_batches_tracked")


class NativeVQAE:
    """Owns a vqae_handle built from a {state-dict name: tensor} mapping in the reference's naming
    (SURVEY.md §5).  One handle per process/device; not thread-safe."""

    def __init__(self, spec: VQAESpec, state_dict, compute_dtype=None):
        """compute_dtype: None/'f32', or 'bf16' / 'f16' (torch dtypes accepted): torch.autocast semantics
        for the convolutions (16-bit operands and conv outputs, fp32 accumulation, fp32 everything else)."""
        self.spec = spec
        self.compute_dtype = L.dtype_code(compute_dtype)
        cfg = L.Config(spec.in_channels, spec.stem, spec.n_down, spec.n_pre, spec.n_post, spec.n_enc,
                       spec.num_embeddings, spec.projection_dim, float(spec.commitment_cost), self.compute_dtype,
                       L.BLOCK_MBCONV if spec.block == "mbconv" else L.BLOCK_FIXUP, spec.expand_ratio,
                       spec.se_divisor, float(spec.bn_eps))
        self._state, items = {}, []             # host copies (fp32): siblings in other compute dtypes are built from them
        for name, t in state_dict.items():
            if name.endswith(_BUFFER_ONLY):
                continue
            a = np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t),
                                     dtype=np.float32)
            self._state[name] = a
            items.append(L.Tensor(name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size))
        self._siblings = {}
        arr = (L.Tensor * len(items))(*items)
        h = ctypes.c_void_p()
        L.check(L.lib().vqae_create(ctypes.byref(cfg), arr, len(items), ctypes.byref(h)))
        self._h = h
        self.channels = spec.channels
        self.factor = 2 ** spec.n_down

    def with_dtype(self, compute_dtype):
        """The handle of the same weights with another conv compute dtype (None / 'f32' / 'bf16' / 'f16' / torch
        dtypes), built on first use and cached: `with torch.autocast('cuda', dtype)` of the reference's run_eval
        (extract_embeddings.py:124-125) maps to `nat.with_dtype(dtype)`."""
        code = L.dtype_code(compute_dtype)
        if code == self.compute_dtype:
            return self
        if self.spec.block == "mbconv" and code != L.DT_F32:
            # The MBConv kernels fold the BatchNorms into the conv weights and run in fp32 only.  Under the reference's autocast the
            # extraction is a 16-bit evaluation that agrees with fp32 on ~99.5 % of the indices: the fp32
            # handle answers instead -- the more accurate of the two evaluations -- rather than failing the default run_eval.
            import warnings
            warnings.warn("vqae_amd: MBConv models run in fp32; the requested 16-bit autocast is ignored", stacklevel=2)
            return self if self.compute_dtype == L.DT_F32 else self.with_dtype(None)
        if code not in self._siblings:
            self._siblings[code] = NativeVQAE(self.spec, self._state, compute_dtype=code)
        return self._siblings[code]

    def close(self):
        for sib in getattr(self, "_siblings", {}).values():
            sib.close()
        self._siblings = {}
        if getattr(self, "_h", None):
            L.lib().vqae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- calls ---------------------------------------------------------------------------------
    def reserve(self, max_batch, h, w):
        L.check(L.lib().vqae_reserve(self._h, max_batch, h, w))

    def set_codebook(self, embed):
        a = np.ascontiguousarray(embed.detach().cpu().numpy() if isinstance(embed, torch.Tensor) else embed,
                                 dtype=np.float32)
        assert a.shape == (self.spec.num_embeddings, self.spec.code_dim), a.shape
        L.check(L.lib().vqae_set_codebook(self._h, a.ctypes.data_as(ctypes.c_void_p)))
        self._state["encoder.vq_layers.0.embed"] = a.copy()
        for sib in self._siblings.values():
            sib.set_codebook(a)

    @staticmethod
    def _layout(x, layout):
        if layout == "NCHW":
            B, _, H, W = x.shape
            return L.LAYOUT_NCHW, B, H, W
        B, H, W, _ = x.shape
        return L.LAYOUT_NHWC, B, H, W

    def encode(self, x, layout="NCHW", idx_dtype=torch.int64, want_q=True, want_loss=True):
        """Encoder.forward: x [B,3,H,W] -> (q [B,C,h,w] | None, idx [B,h,w], loss 0-d | None)."""
        ops._need_gpu(x)
        x = x.contiguous()
        lay, B, H, W = self._layout(x, layout)
        zh, zw = H // self.factor, W // self.factor
        idx = torch.empty((B, zh, zw), dtype=idx_dtype, device=x.device)
        q = None
        if want_q:
            shape = (B, self.channels, zh, zw) if layout == "NCHW" else (B, zh, zw, self.channels)
            q = torch.empty(shape, dtype=torch.float32, device=x.device)
        loss = torch.zeros((), dtype=torch.float32, device=x.device) if want_loss else None
        L.check(L.lib().vqae_encode(self._h, ops._p(x), B, H, W, lay, ops._p(idx), ops.idx_code(idx_dtype),
                                    ops._p(q), ops._p(loss), ops._stream()))
        return q, idx, loss

    def encode_u8(self, x_u8, idx_dtype=torch.int64, want_q=False, q_layout="NCHW"):
        """uint8 NHWC patches [B,H,W,3], normalised on device, -> (q | None, idx, loss)."""
        ops._need_gpu(x_u8)
        assert x_u8.dtype == torch.uint8 and x_u8.shape[-1] == 3
        x_u8 = x_u8.contiguous()
        B, H, W, _ = x_u8.shape
        zh, zw = H // self.factor, W // self.factor
        idx = torch.empty((B, zh, zw), dtype=idx_dtype, device=x_u8.device)
        q = None
        if want_q:
            shape = (B, self.channels, zh, zw) if q_layout == "NCHW" else (B, zh, zw, self.channels)
            q = torch.empty(shape, dtype=torch.float32, device=x_u8.device)
        loss = torch.zeros((), dtype=torch.float32, device=x_u8.device)
        L.check(L.lib().vqae_encode_u8(self._h, ops._p(x_u8), B, H, W, ops._p(idx), ops.idx_code(idx_dtype), ops._p(q),
                                       L.LAYOUT_NCHW if q_layout == "NCHW" else L.LAYOUT_NHWC, ops._p(loss),
                                       ops._stream()))
        return q, idx, loss

    def encode_features(self, x, layout="NCHW"):
        """Pre-VQ activations, NHWC [B,h,w,D] (projected if the model projects)."""
        ops._need_gpu(x)
        x = x.contiguous()
        lay, B, H, W = self._layout(x, layout)
        z = torch.empty((B, H // self.factor, W // self.factor, self.spec.code_dim), dtype=torch.float32,
                        device=x.device)
        L.check(L.lib().vqae_encode_features(self._h, ops._p(x), B, H, W, lay, ops._p(z), ops._stream()))
        return z

    def decode(self, q, layout="NCHW"):
        """Decoder.forward: q [B,C,h,w] -> out [B,3,H,W]."""
        ops._need_gpu(q)
        q = q.contiguous()
        lay, B, qh, qw = self._layout(q, layout)
        H, W = qh * self.factor, qw * self.factor
        shape = (B, self.spec.in_channels, H, W) if layout == "NCHW" else (B, H, W, self.spec.in_channels)
        out = torch.empty(shape, dtype=torch.float32, device=q.device)
        L.check(L.lib().vqae_decode(self._h, ops._p(q), B, qh, qw, lay, ops._p(out), ops._stream()))
        return out

    def decode_indices(self, idx, layout="NCHW"):
        ops._need_gpu(idx)
        idx = idx.contiguous()
        B, qh, qw = idx.shape
        H, W = qh * self.factor, qw * self.factor
        shape = (B, self.spec.in_channels, H, W) if layout == "NCHW" else (B, H, W, self.spec.in_channels)
        out = torch.empty(shape, dtype=torch.float32, device=idx.device)
        L.check(L.lib().vqae_decode_indices(self._h, ops._p(idx), ops.idx_code(idx.dtype), B, qh, qw,
                                            L.LAYOUT_NCHW if layout == "NCHW" else L.LAYOUT_NHWC, ops._p(out),
                                            ops._stream()))
        return out

    def forward(self, x, layout="NCHW", idx_dtype=torch.int64, want_idx=True):
        """VQAE.forward: x -> (out, idx | None, loss)."""
        ops._need_gpu(x)
        x = x.contiguous()
        lay, B, H, W = self._layout(x, layout)
        out = torch.empty_like(x)
        idx = torch.empty((B, H // self.factor, W // self.factor), dtype=idx_dtype, device=x.device) if want_idx else None
        loss = torch.zeros((), dtype=torch.float32, device=x.device)
        L.check(L.lib().vqae_forward(self._h, ops._p(x), B, H, W, lay, ops._p(out), ops._p(idx),
                                     ops.idx_code(idx_dtype), ops._p(loss), ops._stream()))
        return out, idx, loss

    def block_count(self, side):
        return int(L.lib().vqae_block_count(self._h, 0 if side in (0, "encoder") else 1))

    def run_blocks(self, side, first, count, x_nhwc):
        """Blocks [first, first + count) of the encoder ('encoder' / 0) or decoder ('decoder' / 1) block list on
        x [B,H,W,cin] (NHWC) through the handle's own kernel dispatch -> y [B,H',W',cout] (NHWC).  The block
        order is spec.encoder_block_names / decoder_block_names."""
        from .spec import decoder_block_names, encoder_block_names
        ops._need_gpu(x_nhwc)
        x = x_nhwc.contiguous().float()
        B, H, W, cin = x.shape
        s = 0 if side in (0, "encoder") else 1
        names = encoder_block_names(self.spec) if s == 0 else decoder_block_names(self.spec)
        assert 0 <= first and count >= 1 and first + count <= len(names), (first, count, len(names))
        assert cin == names[first][2], f"block {names[first][0]} takes {names[first][2]} channels, got {cin}"
        Ho, Wo = H, W
        for _, mode, _, _ in names[first:first + count]:
            Ho, Wo = (Ho // 2, Wo // 2) if mode == "down" else ((2 * Ho, 2 * Wo) if mode == "up" else (Ho, Wo))
        y = torch.empty((B, Ho, Wo, names[first + count - 1][3]), dtype=torch.float32, device=x.device)
        oh, ow = ctypes.c_int(0), ctypes.c_int(0)
        L.check(L.lib().vqae_run_blocks(self._h, s, first, count, ops._p(x), B, H, W, ops._p(y), ctypes.byref(oh),
                                        ctypes.byref(ow), ops._stream()))
        assert B == 0 or (oh.value, ow.value) == (Ho, Wo)
        return y

    def flops_per_patch(self, h, w, encoder=True, decoder=True):
        return float(L.lib().vqae_flops_per_patch(self._h, h, w, int(encoder), int(decoder)))

    def calibrate_codebook(self, x, embed, layout="NCHW"):
        """embed <- embed * std + mean of the pre-VQ activations of a calibration batch, as the
        reference's `_init_ema` does on the first training batch (vq.py:76-94)."""
        z = self.encode_features(x, layout).reshape(-1, self.spec.code_dim)
        new = embed.to(z.device) * z.std(dim=0) + z.mean(dim=0)
        self.set_codebook(new)
        return new
