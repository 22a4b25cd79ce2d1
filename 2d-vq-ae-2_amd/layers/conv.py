"""Mirror of vq_ae.layers.conv.ResizeConv2D (reference vq_ae/layers/conv.py:4-11):
conv(bicubic_x2(x)), used with kernel_size 1 (conv_layer/up2dresize.yaml)."""
import torch
from torch import nn

from .. import ops


class ResizeConv2D(nn.Conv2d):
    def __init__(self, *conv_args, **conv_kwargs):
        super().__init__(*conv_args, **conv_kwargs)
        if self.kernel_size != (1, 1) or self.stride != (1, 1) or self.padding not in ((0, 0), 0):
            raise NotImplementedError("ResizeConv2D: only the 1x1 form (up2dresize.yaml) is implemented")
        self._packed = None

    def forward(self, data):
        key = (self.weight._version, self.weight.device)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.pack_conv_weight(self.weight.detach()))
        with torch.no_grad():
            x = ops.bicubic_up2(ops.nchw_to_nhwc(data.detach().float()))
            y = ops.conv2d(x, self._packed[1], self.out_channels, 1,
                           bias_vec=self.bias.detach() if self.bias is not None else None)
            return ops.nhwc_to_nchw(y)
