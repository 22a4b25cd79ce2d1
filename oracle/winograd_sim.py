"""TEST INFRASTRUCTURE (like everything under oracle/): a CPU simulation of the Winograd forms the fp32 trunk kernels use for the 3x3
circular conv2 of a Fixup block (reference: vq_ae/layers/conv_block.py:203 through pre_activation_fixup.yaml:56-58) --
F(2x2,3x3) (csrc/conv_wino.hip) and F(4x4,3x3) (csrc/conv_wino43.hip) -- in the dtype of its input.  It exists to MEASURE what the
transforms do to fp32 rounding (tests/test_wino43_math.py, tools/dbg/wino43_numerics.py); the product never
imports it.  Matrices: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (interpolation points 0, +-1, +-2, inf)."""
import torch

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd_conv3x3_circular(x: torch.Tensor, w: torch.Tensor, m: int) -> torch.Tensor:
    """conv2d(circular_pad(x, 1), w) evaluated as F(m x m, 3x3), m in {2, 4}; H and W multiples of m.  The weight transform
    G g G^T is evaluated in fp64 and rounded once (as the kernels' weight-packing launches do); everything else runs in x.dtype."""
    BT, G, AT = (BT4, G4, AT4) if m == 4 else (BT2, G2, AT2)
    t = m + 2
    dt = x.dtype
    U = (G @ w.double() @ G.T).to(dt)                                 # [O, C, t, t]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode="circular")
    d = xp.unfold(2, t, m).unfold(3, t, m)                            # [B, C, th, tw, t, t]
    BTf, ATf = BT.to(dt), AT.to(dt)
    V = torch.einsum("ij,bcyxjk->bcyxik", BTf, d)
    V = torch.einsum("bcyxik,lk->bcyxil", V, BTf)
    B_, C, th, tw = V.shape[:4]
    Vp = V.permute(4, 5, 1, 0, 2, 3).reshape(t * t, C, -1)            # [pos, C, n]
    Up = U.permute(2, 3, 0, 1).reshape(t * t, U.shape[0], C)          # [pos, O, C]
    M = torch.bmm(Up, Vp).reshape(t, t, U.shape[0], B_, th, tw)
    Y = torch.einsum("ai,ijobyx->ajobyx", ATf, M)
    Y = torch.einsum("ajobyx,cj->acobyx", Y, ATf)                     # [m, m, O, B, th, tw]
    return Y.permute(3, 2, 4, 0, 5, 1).reshape(B_, U.shape[0], th * m, tw * m)
