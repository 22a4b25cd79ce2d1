"""Mirror of vq_ae.layers.misc.SELayer (reference vq_ae/layers/misc.py:7-30)."""
import torch
from torch import nn

from .. import ops


def make_divisible(value, divisor, divide=True, min_value=None):
    """utils/train_helpers.py:11-24."""
    return max(divisor if min_value is None else min_value, int(value + divisor / 2)) // (divisor if divide else 1)


class SELayer(nn.Module):
    """Squeeze-excite: x * sigmoid(fc2(SiLU(fc0(mean_hw(x))))).  Parameter names `fc.0.*`, `fc.2.*` as in the
    reference.  Inside MBConv the spatial mean comes from the depthwise kernel's strip sums and the product is
    applied while the next 1x1 conv loads its input; standalone `forward` runs the same kernels."""

    def __init__(self, in_channels: int, out_channels: int, bottleneck_divisor: int):
        super().__init__()
        mid = make_divisible(in_channels, bottleneck_divisor, divide=True)
        self.fc = nn.Sequential(nn.Linear(in_channels, mid), nn.SiLU(), nn.Linear(mid, out_channels), nn.Sigmoid())
        self.in_channels, self.out_channels = in_channels, out_channels

    def gate_from_partial(self, partial, batch, h, w):
        return ops.se_gate(partial, batch, h, w, self.fc[0].weight.detach(), self.fc[0].bias.detach(),
                           self.fc[2].weight.detach(), self.fc[2].bias.detach())

    def forward(self, x: torch.Tensor):
        with torch.no_grad():
            xh = ops.nchw_to_nhwc(x.detach().float())
            B, H, W, C = xh.shape
            # identity depthwise "conv" (one centre tap = 1) only to obtain the strip sums from the same kernel
            taps = torch.zeros((9, C), dtype=torch.float32, device=x.device)
            taps[4] = 1.0
            _, part = ops.dwconv(xh, taps, None, want_partial=True)
            g = self.gate_from_partial(part, B, H, W)
            return x * g.view(B, C, 1, 1)
