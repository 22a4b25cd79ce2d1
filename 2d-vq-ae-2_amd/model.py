"""Drop-in mirrors of vq_ae.model.{Encoder, Decoder, VQAE} (reference vq_ae/model.py:129-217,
220-291, 13-48) whose forward passes run on the native vqae_handle runtime of libvqae_hip.so.

Constructor kwargs are the reference's (conf dicts as Hydra passes them with `_recursive_: False`);
module nesting and parameter names match, so reference state_dicts load unchanged
(`encoder.down_layers.0.layers.<lvl>.layers.<blk>.*`).  Only what the shipped
configs compose is implemented: one VQ level, Fixup blocks, no shortcut blocks; anything else raises
NotImplementedError.
"""
import weakref
from typing import Sequence

import torch
from torch import nn

from .layers.conv_block import DownBlock, MBConv, PreActFixupResBlock, UpBlock, _build
from .layers.vq import EMAVectorQuantizer, ProjectedEMAVectorQuantizer2d
from .native import NativeVQAE
from .spec import VQAESpec

_META = ("_target_", "_recursive_", "_convert_", "_partial_")


def _strip(conf):
    return {k: v for k, v in dict(conf).items() if k not in _META}


def _single(x, what):
    if isinstance(x, (list, tuple)):
        if len(x) != 1:
            raise NotImplementedError(f"multi-level VQ-AE ({what} has {len(x)} entries) is not implemented")
        return x[0]
    return x


def _vq_level_confs(vq_conf):
    """The reference passes {'_target_': instantiate_dictified_listconf, '0': {...}} (encoder/default.yaml:10-13)."""
    if isinstance(vq_conf, (list, tuple)):
        return list(vq_conf)
    tgt = str(vq_conf.get("_target_", ""))
    if tgt.endswith("VectorQuantizer") or tgt.endswith("VectorQuantizer2d"):
        return [vq_conf]
    return [v for k, v in vq_conf.items() if k not in _META]


def _make_stem(conf, what):
    c = _strip(conf)
    k = c.get("kernel_size", 3)
    if (k, c.get("stride", 1), c.get("padding", 1), c.get("padding_mode", "zeros"), c.get("bias", True)) != \
            (3, 1, 1, "zeros", True):
        raise NotImplementedError(f"{what}: only Conv2d(k3,s1,p1,zeros,bias) is implemented (same2d.yaml)")
    return nn.Conv2d(c["in_channels"], c["out_channels"], 3, padding=1)


def _block_hp(conv_block_conf):
    """VQAESpec fields that select the conv block family (Fixup | MBConv) from its conf dict."""
    c = dict(conv_block_conf)
    if str(c.get("_target_", "")).endswith("MBConv") or "expand_ratio" in c:
        er = c.get("expand_ratio", 4)
        if int(er) != er:
            raise NotImplementedError(f"MBConv expand_ratio {er}: only integer ratios are implemented")
        return dict(block="mbconv", expand_ratio=int(er),
                    se_divisor=int((c.get("se_conf") or {}).get("bottleneck_divisor", 4)),
                    bn_eps=float((c.get("batchnorm_conf") or {}).get("eps", 1e-5)))
    return {}


class _NativeMixin:
    """Lazily builds (and shares) the vqae_handle from the module's own parameters."""
    _native = None
    _prefix = ""

    def _spec(self) -> VQAESpec:
        raise NotImplementedError

    @staticmethod
    def _autocast_dtype():
        """The reference's extraction runs `with torch.autocast('cuda')` (extract_embeddings.py:124-125);
        inside such a context the mirrors use the matching 16-bit autocast handle."""
        if torch.is_autocast_enabled("cuda"):
            return torch.get_autocast_dtype("cuda")
        return None

    def _weights_signature(self):
        """Changes whenever a parameter / buffer is modified in place (optimizer step, `.data.copy_`, the VQ's EMA
        update) or replaced: the device snapshot is then rebuilt on the next call instead of silently going stale."""
        return tuple((id(t), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def native(self) -> NativeVQAE:
        owner = getattr(self, "_owner", None)
        if owner is not None and owner() is not None:
            return owner().native()
        dt = self._autocast_dtype()
        if dt is not None and self._spec().block == "mbconv":          # MBConv kernels are fp32 only: see NativeVQAE.with_dtype
            import warnings
            warnings.warn("vqae_amd: MBConv models run in fp32; the surrounding torch.autocast is ignored", stacklevel=3)
            dt = None
        sig = self._weights_signature()
        if self._native is None or self._native.get("sig") != sig:
            for k, v in (self._native or {}).items():
                if k != "sig":
                    v.close()
            self._native = {"sig": sig}
        if dt not in self._native:
            sd = {self._prefix + k: v for k, v in self.state_dict().items()}
            self._native[dt] = NativeVQAE(self._spec(), sd, compute_dtype=dt)
        return self._native[dt]

    @staticmethod
    def _inference_only(*tensors):
        """These mirrors are inference-only (no autograd through the HIP path; SURVEY.md section 2 scopes training out):
        fail loudly instead of returning values that would train to nothing."""
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
            raise NotImplementedError("vqae_amd.model mirrors run the HIP inference path: inputs that require grad are not "
                                      "supported (wrap the call in torch.no_grad())")

    def refresh(self):
        """Drop the device snapshot of the weights (call after changing parameters)."""
        self._native = None
        for m in self.children():
            if isinstance(m, _NativeMixin):
                m.refresh()

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.refresh()
        return r

    def __getstate__(self):
        """copy.deepcopy / pickle: the device snapshot (a raw vqae_handle) and the back-reference to the parent VQAE are
        per-object state -- a copy builds its own handle from its own parameters on first use, and a copied VQAE rebinds
        its children to itself (VQAE.__setstate__)."""
        d = self.__dict__.copy()
        d.pop("_owner", None)
        d["_native"] = None
        return d


class Encoder(_NativeMixin, nn.Module):
    _prefix = "encoder."

    def __init__(self, stem_conf, down_block_conf, n_pre_enc_layers, vq_conf, conv_block_conf,
                 shortcut_block_conf=None):
        super().__init__()
        if shortcut_block_conf:
            raise NotImplementedError("shortcut blocks (multi-level VQ) are not implemented")
        vq_levels = _vq_level_confs(vq_conf)
        if len(vq_levels) != 1:
            raise NotImplementedError("only single-level VQ (every shipped config) is implemented")
        down = _strip(_single(down_block_conf, "down_block_conf"))
        n_enc = int(_single(n_pre_enc_layers, "n_pre_enc_layers"))

        self.in_stem = _make_stem(stem_conf, "in_stem")                    # model.py:141
        down.pop("in_channels", None)
        self.down_layers = nn.ModuleList([DownBlock(in_channels=self.in_stem.out_channels, **down)])
        c = self.down_layers[0].out_channels
        self.pre_enc_layers = nn.ModuleList([nn.Sequential(*(
            _build(conv_block_conf, mode="same", in_channels=c, out_channels=c) for _ in range(n_enc)))])
        vq = _strip(vq_levels[0])
        projected = "projection_dim" in vq
        self.vq_layers = nn.ModuleList([(ProjectedEMAVectorQuantizer2d if projected else EMAVectorQuantizer)(**vq)])
        if vq["embedding_dim"] != c:
            raise NotImplementedError(f"VQ embedding_dim {vq['embedding_dim']} != encoder channels {c}")
        self._hp = dict(stem=self.in_stem.out_channels, in_channels=self.in_stem.in_channels, n_down=down["n_down"],
                        n_pre=down.get("n_pre_layers") or 0, n_post=down.get("n_post_layers") or 0, n_enc=n_enc,
                        num_embeddings=vq["num_embeddings"], projection_dim=vq.get("projection_dim", 0),
                        commitment_cost=float(vq["commitment_cost"]), decay=float(vq["decay"]),
                        laplace_alpha=float(vq["laplace_alpha"]), **_block_hp(conv_block_conf))

    def _spec(self):
        return VQAESpec(**self._hp)

    def forward(self, x: torch.Tensor):
        """-> ((q,), (idx,), (loss,)), low-res first (model.py:189-217)."""
        self._inference_only(x)
        q, idx, loss = self.native().encode(x.float(), "NCHW")
        return (q,), (idx,), (loss,)


class Decoder(_NativeMixin, nn.Module):
    _prefix = "decoder."

    def __init__(self, n_enc_layers, stem_conf, up_block_conf, n_post_enc_layers, conv_block_conf,
                 shortcut_block_conf=None):
        super().__init__()
        if shortcut_block_conf or n_enc_layers != 1:
            raise NotImplementedError("only single-level decoders (every shipped config) are implemented")
        up = _strip(_single(up_block_conf, "up_block_conf"))
        n_enc = int(_single(n_post_enc_layers, "n_post_enc_layers"))
        self.out_stem = _make_stem(stem_conf, "out_stem")                  # model.py:232
        up.pop("out_channels", None)
        self.up_layers = nn.ModuleList([UpBlock(out_channels=self.out_stem.in_channels, **up)])
        c = self.up_layers[0].in_channels
        self.post_enc_layers = nn.ModuleList([nn.Sequential(*(
            _build(conv_block_conf, mode="same", in_channels=c, out_channels=c) for _ in range(n_enc)))])
        self._hp = dict(stem=self.out_stem.in_channels, in_channels=self.out_stem.out_channels, n_down=up["n_up"],
                        n_pre=up.get("n_pre_layers") or 0, n_post=up.get("n_post_layers") or 0, n_enc=n_enc,
                        num_embeddings=1, projection_dim=0, **_block_hp(conv_block_conf))

    def _spec(self):
        return VQAESpec(**self._hp)

    def forward(self, x: Sequence[torch.Tensor]) -> torch.Tensor:
        """x: encodings low-res -> high-res (one level) -> reconstruction (model.py:274-291)."""
        if len(x) != 1:
            raise NotImplementedError("only single-level decoders are implemented")
        self._inference_only(x[0])
        return self.native().decode(x[0].float(), "NCHW")


class VQAE(_NativeMixin, nn.Module):
    """VQAE.forward (model.py:41-48).  The Lightning training harness (optimisers, logging) is out of
    scope; `optim_conf` / `loss_f_conf` are accepted and stored for signature compatibility."""

    def __init__(self, optim_conf=None, loss_f_conf=None, encoder_conf=None, decoder_conf=None, **kwargs):
        super().__init__()
        self.optim_conf, self.loss_f_conf = optim_conf, loss_f_conf
        self.encoder = Encoder(**_strip(encoder_conf))
        self.decoder = Decoder(**_strip(decoder_conf))
        self._bind_children()
        for k, v in kwargs.items():
            setattr(self, k, v)

    def _bind_children(self):
        # the children run on the parent's handle (one device copy of the weights; it encodes and decodes)
        object.__setattr__(self.encoder, "_owner", weakref.ref(self))
        object.__setattr__(self.decoder, "_owner", weakref.ref(self))

    def __setstate__(self, state):
        super().__setstate__(state)
        self._bind_children()

    def _spec(self):
        return self.encoder._spec()


    def forward(self, data: torch.Tensor):
        self._inference_only(data)
        out, _, loss = self.native().forward(data.float(), "NCHW", want_idx=False)
        return out, (loss,)

    # ---- convenience constructors --------------------------------------------------------------
    @classmethod
    def from_spec(cls, spec: VQAESpec):
        return cls(**default_confs(spec))


def default_confs(spec: VQAESpec):
    """The nested conf dicts Hydra composes from conf/model/vq_ae.yaml for `spec`
    (SURVEY.md Appendix A), with this package's classes as `_target_`s."""
    def conv(k, stride=1, padding=0, bias=True, padding_mode="zeros"):
        return {"in_channels": None, "out_channels": None, "kernel_size": k, "stride": stride, "padding": padding,
                "dilation": 1, "groups": 1, "bias": bias, "padding_mode": padding_mode}
    proj = lambda: conv(1, bias=False)
    n_layers = (spec.n_down * spec.n_pre * spec.n_post) * 2 + 2 * spec.n_enc     # n_layers.yaml:3
    circ3 = lambda: conv(3, 1, 1, False, "circular")
    down2 = lambda: conv(2, 2, bias=False, padding_mode="circular")
    up2 = lambda: {"in_channels": None, "out_channels": None, "kernel_size": 2, "stride": 2, "padding": 0,
                   "output_padding": 0, "groups": 1, "bias": False, "dilation": 1, "padding_mode": "zeros"}
    mb = {"_target_": "vqae_amd.layers.conv_block.MBConv", "_recursive_": False, "expand_ratio": spec.expand_ratio,
          "activation_conf": {"_target_": "torch.nn.SiLU"},
          "se_conf": {"_target_": "vqae_amd.layers.misc.SELayer", "bottleneck_divisor": spec.se_divisor},
          "batchnorm_conf": {"_target_": "torch.nn.BatchNorm2d", "eps": spec.bn_eps, "momentum": 0.1, "affine": True,
                             "track_running_stats": True},
          "conv_conf": {
              "down": {"branch_conv1": proj(), "branch_conv2": down2(), "branch_conv3": proj(), "skip_conv": down2()},
              "up": {"branch_conv1": proj(), "branch_conv2": up2(), "branch_conv3": proj(), "skip_conv": up2()},
              "same": {"branch_conv1": proj(), "branch_conv2": circ3(), "branch_conv3": proj(), "skip_conv": proj()},
              "out": {"branch_conv1": proj(), "branch_conv2": circ3(), "branch_conv3": proj(), "skip_conv": circ3()}}}
    fx = {"_target_": "vqae_amd.layers.conv_block.PreActFixupResBlock", "_recursive_": False,
          "bottleneck_divisor": 1, "n_layers": n_layers,
          "activation": {"_target_": "torch.nn.ELU", "alpha": 1.0},
          "conv_conf": {
              "down": {"branch_conv1": proj(), "branch_conv2": conv(2, 2, bias=False, padding_mode="circular"),
                       "branch_conv3": proj(), "skip_conv": conv(2, 2, bias=False, padding_mode="circular")},
              "up": {"branch_conv1": proj(), "branch_conv2": proj(), "branch_conv3": proj(), "skip_conv": proj()},
              "same": {"branch_conv1": proj(), "branch_conv2": conv(3, 1, 1, False, "circular"),
                       "branch_conv3": proj(), "skip_conv": proj()}}}
    if spec.block == "mbconv":                   # conf/model/{encoder,decoder}/efficientnetv2.yaml
        fx = mb
    vq = {"num_embeddings": spec.num_embeddings, "embedding_dim": spec.channels,
          "commitment_cost": spec.commitment_cost, "decay": spec.decay, "laplace_alpha": spec.laplace_alpha}
    if spec.projection_dim > 0:
        vq.update(_target_="vqae_amd.layers.vq.ProjectedEMAVectorQuantizer2d", projection_dim=spec.projection_dim)
    else:
        vq.update(_target_="vqae_amd.layers.vq.EMAVectorQuantizer")
    stem_in, stem_out = conv(3, padding=1), conv(3, padding=1)
    stem_in.update(in_channels=spec.in_channels, out_channels=spec.stem)
    stem_out.update(in_channels=spec.stem, out_channels=spec.in_channels)
    enc = {"_target_": "vqae_amd.model.Encoder", "_recursive_": False, "stem_conf": stem_in,
           "down_block_conf": {"_target_": "vqae_amd.layers.conv_block.DownBlock", "_recursive_": False,
                               "n_down": spec.n_down, "n_pre_layers": spec.n_pre, "n_post_layers": spec.n_post,
                               "conv_conf": fx},
           "n_pre_enc_layers": spec.n_enc, "vq_conf": {"0": vq}, "conv_block_conf": fx, "shortcut_block_conf": None}
    dec = {"_target_": "vqae_amd.model.Decoder", "_recursive_": False, "n_enc_layers": 1, "stem_conf": stem_out,
           "up_block_conf": {"_target_": "vqae_amd.layers.conv_block.UpBlock", "_recursive_": False,
                             "n_up": spec.n_down, "n_pre_layers": spec.n_pre, "n_post_layers": spec.n_post,
                             "conv_conf": fx},
           "n_post_enc_layers": spec.n_enc, "conv_block_conf": fx, "shortcut_block_conf": None}
    return {"optim_conf": None, "loss_f_conf": None, "encoder_conf": enc, "decoder_conf": dec}


def load_lightning_state_dict(path):
    """state_dict of a Lightning checkpoint written by the reference's training run
    (`VQAE.load_from_checkpoint`, extract_embeddings.py:156).  Only tensors are read
    (`weights_only=True`: nothing in the file is executed); the pickled OmegaConf hyper-parameters are
    NOT loaded -- build the model from a VQAESpec / conf dicts and pass the result to `load_state_dict`."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}
