"""Drop-in mirrors of vq_ae.layers.conv_block.{PreActFixupResBlock, EnvelopBlock, DownBlock, UpBlock}
(reference vq_ae/layers/conv_block.py:18-237) backed by libvqae_hip.so.

Constructor kwargs, parameter names (bias1a..bias4, scale, branch_conv{1,2,3}.weight,
skip_conv.weight, bias1c/bias1d) and module nesting (`layers.<i>`) match the reference, so its
state_dicts load unchanged.  The conf dicts that Hydra passes (`conv_conf`, `activation`,
`_recursive_: False`) are validated against what the HIP kernels implement -- the default Fixup
configuration of conf/model/layers/conv_block/pre_activation_fixup.yaml -- and anything else raises
NotImplementedError instead of silently computing something different.
"""
from math import isclose
from typing import Optional

import numpy as np
import torch
from torch import nn

from .. import _lib as L
from .. import ops

_MODES = ("down", "same", "up", "out")


def _conv_kind(conf, default):
    """(ksize, stride, pad, circular) of a conv_layer conf dict, or `default` when conf is None."""
    if conf is None:
        return default
    k = conf.get("kernel_size", default[0])
    k = k[0] if isinstance(k, (tuple, list)) else k
    s = conf.get("stride", 1)
    s = s[0] if isinstance(s, (tuple, list)) else s
    p = conf.get("padding", 0)
    p = p[0] if isinstance(p, (tuple, list)) else p
    if conf.get("bias", False):
        raise NotImplementedError("Fixup block convs with bias are not implemented (reference conf: bias False)")
    if conf.get("groups", 1) not in (1, None) or conf.get("dilation", 1) != 1:
        raise NotImplementedError("grouped / dilated convs are not implemented")
    return int(k), int(s), int(p), conf.get("padding_mode", "zeros") == "circular"


class _Weight(nn.Module):
    """Parameter holder so state-dict keys read `<name>.weight` like the reference's nn.Conv2d."""

    def __init__(self, cout, cin, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.out_channels, self.in_channels, self.kernel_size = cout, cin, (k, k)
        self._packed = None

    def packed(self):
        key = (self.weight._version, self.weight.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.pack_conv_weight(self.weight.detach()))
        return self._packed[1]


class PreActFixupResBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, mode: str, bottleneck_divisor: float = 1,
                 activation=None, conv_conf=None, n_layers: Optional[int] = None):
        super().__init__()
        assert mode in _MODES                                              # conv_block.py:148
        max_channels = max(in_channels, out_channels)
        assert isclose(max_channels % bottleneck_divisor, 0), (          # conv_block.py:152-154
            f"residual channels: {max_channels} not divisible by bottleneck divisor: {bottleneck_divisor}!")
        branch = max(round(max_channels / bottleneck_divisor), 1)
        if activation is not None:
            tgt = activation.get("_target_", "torch.nn.ELU") if isinstance(activation, dict) else type(activation).__name__
            alpha = activation.get("alpha", 1.0) if isinstance(activation, dict) else getattr(activation, "alpha", 1.0)
            if not str(tgt).endswith("ELU") or float(alpha) != 1.0:
                raise NotImplementedError(f"only ELU(alpha=1) is implemented (activation/elu.yaml); got {tgt}")
        cc = (conv_conf or {}).get(mode, {}) if conv_conf is not None else {}
        # branch_conv2: same2d / out2d (3x3 s1 p1, circular by the block conf), down2d (2x2 s2), up2dresize (bicubic x2 + 1x1)
        want2 = {"same": (3, 1, 1, True), "out": (3, 1, 1, True), "down": (2, 2, 0, False), "up": (1, 1, 0, False)}[mode]
        got2 = _conv_kind(cc.get("branch_conv2"), want2)
        if got2[:3] != want2[:3] or (mode in ("same", "out") and not got2[3]):
            raise NotImplementedError(f"branch_conv2 {got2} for mode '{mode}' is not the implemented {want2}")
        for nm in ("branch_conv1", "branch_conv3"):
            if _conv_kind(cc.get(nm), (1, 1, 0, False))[:3] != (1, 1, 0):
                raise NotImplementedError(f"{nm} must be a 1x1 conv (proj2d.yaml)")
        self.mode, self.in_channels, self.out_channels, self.branch_channels = mode, in_channels, out_channels, branch

        self.bias1a, self.bias1b, self.bias2a, self.bias2b, self.bias3a, self.bias3b, self.bias4 = (
            nn.Parameter(torch.zeros(1)) for _ in range(7))
        self.scale = nn.Parameter(torch.ones(1))
        self.branch_conv1 = _Weight(branch, in_channels, 1)
        self.branch_conv2 = _Weight(branch, branch, want2[0])
        self.branch_conv3 = _Weight(out_channels, branch, 1)
        if not (mode in ("same", "out") and in_channels == out_channels):  # conv_block.py:180-191
            # skip_conv: down2d (2x2 s2) / up2dresize (1x1 after the resize) / proj2d (1x1, 'same') / out2d (3x3 s1 p1 ZERO padding:
            # pre_activation_fixup.yaml sets padding_mode 'circular' for out.branch_conv2 only)
            want_s = {"down": (2, 2, 0), "up": (1, 1, 0), "same": (1, 1, 0), "out": (3, 1, 1)}[mode]
            got_s = _conv_kind(cc.get("skip_conv"), want_s + (False,))
            if got_s[:3] != want_s or (mode == "out" and got_s[3]):
                raise NotImplementedError(f"skip_conv {got_s} for mode '{mode}' is not the implemented {want_s} (zero padding)")
            self.bias1c, self.bias1d = (nn.Parameter(torch.zeros(1)) for _ in range(2))
            self.skip_conv = _Weight(out_channels, in_channels, want_s[0])
        else:
            self.skip_conv = None
        if n_layers is not None:
            self.initialize_weights(n_layers)

    @torch.no_grad()
    def initialize_weights(self, num_layers):                               # conv_block.py:218-237
        w = self.branch_conv1.weight
        nn.init.normal_(w, mean=0, std=np.sqrt(2 / (w.shape[0] * np.prod(w.shape[2:]))) * num_layers ** (-0.5))
        nn.init.kaiming_normal_(self.branch_conv2.weight)
        nn.init.constant_(self.branch_conv3.weight, val=0)
        if self.skip_conv is not None:
            nn.init.xavier_normal_(self.skip_conv.weight)

    def forward_nhwc(self, x):
        """conv_block.py:196-216 on an NHWC tensor (the block's native layout)."""
        f = lambda p: float(p.detach())
        br = self.branch_channels
        if (self.mode in ("same", "out") and self.skip_conv is None and br == self.in_channels
                and ops.fixup_same_supported(self.in_channels, x.shape[1], x.shape[2])):
            return ops.fixup_same_block(x, self.branch_conv1.packed(), self.branch_conv2.packed(),
                                        self.branch_conv3.packed(),
                                        [f(self.bias1a), f(self.bias1b), f(self.bias2a), f(self.bias2b),
                                         f(self.bias3a), f(self.bias3b), f(self.bias4), f(self.scale)])
        t = ops.conv2d(x, self.branch_conv1.packed(), br, 1, pre=(f(self.bias1a), f(self.bias1b)),
                       act=(f(self.bias2a), f(self.bias2b)))
        if self.mode in ("same", "out"):
            t = ops.conv2d(t, self.branch_conv2.packed(), br, 3, 1, 1, L.PAD_CIRCULAR,
                           act=(f(self.bias3a), f(self.bias3b)))
            if self.skip_conv is None:
                skip = x
            elif self.mode == "same":                                       # proj2d: skip_conv(inp + bias1c) + bias1d
                skip = ops.conv2d(x, self.skip_conv.packed(), self.out_channels, 1, pre=(f(self.bias1c),),
                                  bias_s=f(self.bias1d))
            else:                                                           # out2d: 3x3, zero padding
                skip = ops.conv2d(x, self.skip_conv.packed(), self.out_channels, 3, 1, 1, L.PAD_ZEROS,
                                  pre=(f(self.bias1c),), bias_s=f(self.bias1d))
        elif self.mode == "down":
            t = ops.conv2d(t, self.branch_conv2.packed(), br, 2, 2, 0, act=(f(self.bias3a), f(self.bias3b)))
            skip = ops.conv2d(x, self.skip_conv.packed(), self.out_channels, 2, 2, 0, pre=(f(self.bias1c),),
                              bias_s=f(self.bias1d))
        else:
            t = ops.conv2d(ops.bicubic_up2(t), self.branch_conv2.packed(), br, 1, act=(f(self.bias3a), f(self.bias3b)))
            skip = ops.conv2d(ops.bicubic_up2(x, f(self.bias1c)), self.skip_conv.packed(), self.out_channels, 1,
                              bias_s=f(self.bias1d))
        return ops.conv2d(t, self.branch_conv3.packed(), self.out_channels, 1,
                          scale_bias=(f(self.scale), f(self.bias4)), residual=skip)

    def forward(self, inp: torch.Tensor):
        with torch.no_grad():
            return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(inp.detach().float())))


class MBConv(nn.Module):
    """Mirror of vq_ae.layers.conv_block.MBConv (conv_block.py:240-321), inference mode (BatchNorm running
    statistics): `branch` = [0 1x1, 1 BN, 2 SiLU, 3 depthwise, 4 BN, 5 SiLU, 6 SELayer, 7 1x1, 8 BN] + `skip_conv`,
    same state-dict names.  Only the shipped conf (mbconv.yaml: SiLU, BatchNorm2d, SELayer, bias-free convs) is
    implemented; training-mode batch statistics are not."""

    def __init__(self, in_channels: int, out_channels: int, mode: str, expand_ratio: float, activation_conf=None,
                 conv_conf=None, batchnorm_conf=None, se_conf=None):
        super().__init__()
        from .misc import SELayer
        assert mode in ("down", "same", "up", "out")                        # conv_block.py:254
        max_channels = max(in_channels, out_channels)
        assert isclose(max_channels * expand_ratio % 1, 0), (              # conv_block.py:256-258
            f"max_channels: {max_channels} x expand_ratio: {expand_ratio} % 1 !≈ 0!")
        e = round(max_channels * expand_ratio)
        if activation_conf is not None and not str(activation_conf.get("_target_", "SiLU")).endswith("SiLU"):
            raise NotImplementedError("MBConv: only SiLU is implemented (activation/silu.yaml)")
        if batchnorm_conf is None or se_conf is None:
            raise NotImplementedError("MBConv without BatchNorm / SELayer is not implemented (mbconv.yaml has both)")
        bn_kw = {k: v for k, v in dict(batchnorm_conf).items() if k in ("eps", "momentum", "affine", "track_running_stats")}
        if not bn_kw.get("affine", True) or not bn_kw.get("track_running_stats", True):
            raise NotImplementedError("MBConv: BatchNorm2d must be affine with running statistics (batchnorm2d.yaml)")
        cc = (conv_conf or {}).get(mode, {}) if conv_conf is not None else {}
        for nm in ("branch_conv1", "branch_conv2", "branch_conv3", "skip_conv"):
            if (cc.get(nm) or {}).get("bias", False):
                raise NotImplementedError(f"MBConv: {nm} with bias is not implemented (mbconv.yaml: bias False)")
        k2 = {"same": 3, "out": 3, "down": 2, "up": 2}[mode]
        self.mode, self.in_channels, self.out_channels, self.expanded = mode, in_channels, out_channels, e
        self.branch = nn.Sequential(
            _Weight(e, in_channels, 1), nn.BatchNorm2d(e, **bn_kw), nn.SiLU(),
            _Weight(e, 1, k2), nn.BatchNorm2d(e, **bn_kw), nn.SiLU(),
            SELayer(e, e, int(se_conf.get("bottleneck_divisor", 4))),
            _Weight(out_channels, e, 1), nn.BatchNorm2d(out_channels, **bn_kw))
        if not (mode in ("same", "out") and in_channels == out_channels):  # conv_block.py:303-310
            if mode == "up":                                                # ConvTranspose2d weight: [in][out][k][k]
                self.skip_conv = _Weight(in_channels, out_channels, 2)
            else:
                self.skip_conv = _Weight(out_channels, in_channels, {"same": 1, "out": 3, "down": 2}[mode])
        else:
            self.skip_conv = None
        with torch.no_grad():                                               # init batchnorm gamma to 0, :312-314
            self.branch[-1].weight *= 0
        self._prep = None

    @staticmethod
    def _fold(bn):
        g = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
        return g, bn.bias.detach() - bn.running_mean.detach() * g

    def _prepared(self):
        ts = [t for t in list(self.parameters()) + list(self.buffers())]
        key = tuple((t._version, t.device) for t in ts)
        if self._prep is None or self._prep[0] != key:
            b = self.branch
            g1, s1 = self._fold(b[1]); g2, s2 = self._fold(b[4]); g3, s3 = self._fold(b[8])
            w1 = ops.pack_conv_weight((b[0].weight.detach() * g1.view(-1, 1, 1, 1)).contiguous())
            e = self.expanded
            taps = (b[3].weight.detach().reshape(e, -1) * g2.view(-1, 1)).t().contiguous()       # [k*k][e]
            w3 = ops.pack_conv_weight((b[7].weight.detach() * g3.view(-1, 1, 1, 1)).contiguous())
            sk = None
            if self.skip_conv is not None:
                w = self.skip_conv.weight.detach()
                if self.mode == "up":      # [cin][cout][a][b] -> 1x1 conv with outputs ordered (a, b, cout)
                    w = w.permute(2, 3, 1, 0).reshape(4 * self.out_channels, self.in_channels, 1, 1)
                sk = ops.pack_conv_weight(w.contiguous())
            self._prep = (key, dict(w1=w1, s1=s1.contiguous(), taps=taps, s2=s2.contiguous(), w3=w3,
                                    s3=s3.contiguous(), sk=sk))
        return self._prep[1]

    def forward_nhwc(self, x):
        """conv_block.py:316-321 on an NHWC tensor, eval mode."""
        if self.training:
            raise NotImplementedError("MBConv: training-mode BatchNorm is not implemented; call .eval()")
        p = self._prepared()
        B, H, W, _ = x.shape
        e = self.expanded
        t = ops.conv2d(x, p["w1"], e, 1, bias_vec=p["s1"], act="silu")
        dwm = {"same": L.DW_SAME, "out": L.DW_SAME, "down": L.DW_DOWN, "up": L.DW_UP}[self.mode]
        t, part = ops.dwconv(t, p["taps"], p["s2"], dwm, silu=True, want_partial=True)
        gate = self.branch[6].gate_from_partial(part, B, t.shape[1], t.shape[2])
        if self.skip_conv is None:
            skip = x
        elif self.mode == "same":
            skip = ops.conv2d(x, p["sk"], self.out_channels, 1)
        elif self.mode == "out":
            skip = ops.conv2d(x, p["sk"], self.out_channels, 3, 1, 1, L.PAD_CIRCULAR)
        elif self.mode == "down":
            skip = ops.conv2d(x, p["sk"], self.out_channels, 2, 2, 0)
        else:
            skip = ops.pixel_shuffle2(ops.conv2d(x, p["sk"], 4 * self.out_channels, 1), self.out_channels)
        return ops.conv2d(t, p["w3"], self.out_channels, 1, bias_vec=p["s3"], residual=skip, gate=gate)

    def forward(self, inp: torch.Tensor):
        with torch.no_grad():
            return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(inp.detach().float())))


def _build(conf, **kw):
    """Instantiate a block conf dict ({'_target_': ..., kwargs}) with our classes."""
    target = str(dict(conf).get("_target_", ""))
    conf = {k: v for k, v in dict(conf).items() if k not in ("_target_", "_recursive_", "_convert_", "_partial_")}
    conf.update(kw)
    if target.endswith("MBConv") or "expand_ratio" in conf:
        return MBConv(**conf)
    return PreActFixupResBlock(**conf)


class EnvelopBlock(nn.Module):                                              # conv_block.py:94-129
    def __init__(self, envelop_conf, in_channels: int, out_channels: int, pre_layers=None, post_layers=None):
        super().__init__()

        def seq(layers, cin, cout):
            if not layers:
                return []
            if isinstance(layers, (list, tuple)) and len(layers) == 2 and isinstance(layers[1], int):
                layers = [layers[0]] * layers[1]
            elif not isinstance(layers, (list, tuple)):
                layers = [layers]
            return [_build(l, in_channels=cin, out_channels=cout) for l in layers if l]

        self.layers = nn.Sequential(*seq(pre_layers, in_channels, in_channels),
                                    _build(envelop_conf, in_channels=in_channels, out_channels=out_channels),
                                    *seq(post_layers, out_channels, out_channels))

    def forward_nhwc(self, x):
        for l in self.layers:
            x = l.forward_nhwc(x)
        return x

    def forward(self, x):
        with torch.no_grad():
            return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x.detach().float())))


class DownBlock(nn.Module):                                                 # conv_block.py:18-52
    out_channels: int

    def __init__(self, in_channels: int, n_down: int, conv_conf, n_pre_layers: Optional[int],
                 n_post_layers: Optional[int]):
        super().__init__()
        pre, post = ([{**conv_conf, "mode": "same"}] * (n or 0) for n in (n_pre_layers, n_post_layers))
        self.layers = nn.Sequential(*(
            EnvelopBlock({**conv_conf, "mode": "down"}, in_channels * 2 ** j, in_channels * 2 ** (j + 1), pre, post)
            for j in range(n_down)))
        self.out_channels = in_channels * 2 ** n_down

    def forward_nhwc(self, x):
        for l in self.layers:
            x = l.forward_nhwc(x)
        return x

    def forward(self, x):
        with torch.no_grad():
            return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x.detach().float())))


class UpBlock(nn.Module):                                                   # conv_block.py:55-91
    in_channels: int

    def __init__(self, out_channels: int, n_up: int, conv_conf, n_pre_layers: Optional[int],
                 n_post_layers: Optional[int]):
        super().__init__()
        pre, post = ([{**conv_conf, "mode": "same"}] * (n or 0) for n in (n_pre_layers, n_post_layers))
        self.layers = nn.Sequential(*(
            EnvelopBlock({**conv_conf, "mode": "up"}, out_channels * 2 ** (j + 1), out_channels * 2 ** j, pre, post)
            for j in range(n_up - 1, -1, -1)))
        self.in_channels = out_channels * 2 ** n_up

    def forward_nhwc(self, x):
        for l in self.layers:
            x = l.forward_nhwc(x)
        return x

    def forward(self, x):
        with torch.no_grad():
            return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x.detach().float())))
