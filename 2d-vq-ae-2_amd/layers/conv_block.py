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
        if mode == "out":
            raise NotImplementedError("mode 'out' is unused by Encoder/Decoder and not implemented")
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
        want2 = {"same": (3, 1, 1, True), "down": (2, 2, 0, False), "up": (1, 1, 0, False)}[mode]
        got2 = _conv_kind(cc.get("branch_conv2"), want2)
        if got2[:3] != want2[:3] or (mode == "same" and not got2[3]):
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
            if mode == "same":
                raise NotImplementedError("'same' blocks with in_channels != out_channels are not implemented")
            self.bias1c, self.bias1d = (nn.Parameter(torch.zeros(1)) for _ in range(2))
            self.skip_conv = _Weight(out_channels, in_channels, 2 if mode == "down" else 1)
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
        if (self.mode == "same" and br == self.in_channels
                and ops.fixup_same_supported(self.in_channels, x.shape[1], x.shape[2])):
            return ops.fixup_same_block(x, self.branch_conv1.packed(), self.branch_conv2.packed(),
                                        self.branch_conv3.packed(),
                                        [f(self.bias1a), f(self.bias1b), f(self.bias2a), f(self.bias2b),
                                         f(self.bias3a), f(self.bias3b), f(self.bias4), f(self.scale)])
        t = ops.conv2d(x, self.branch_conv1.packed(), br, 1, pre=(f(self.bias1a), f(self.bias1b)),
                       act=(f(self.bias2a), f(self.bias2b)))
        if self.mode == "same":
            t = ops.conv2d(t, self.branch_conv2.packed(), br, 3, 1, 1, L.PAD_CIRCULAR,
                           act=(f(self.bias3a), f(self.bias3b)))
            skip = x
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


def _build(conf, **kw):
    """Instantiate a block conf dict ({'_target_': ..., kwargs}) with our classes."""
    conf = {k: v for k, v in dict(conf).items() if k not in ("_target_", "_recursive_", "_convert_", "_partial_")}
    conf.update(kw)
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
