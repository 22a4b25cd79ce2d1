"""CPU checks of the Winograd forms behind the fp32 trunk kernels (oracle/winograd_sim.py; no GPU): the algebra -- both forms equal the
reference's circular 3x3 conv (conv_block.py:203, pre_activation_fixup.yaml:56-58) in fp64 -- and the size of their fp32 rounding,
which DESIGN.md section 4 quotes: F(4x4,3x3) (csrc/conv_wino43.hip, round 3) is a few times noisier than the direct form and than
F(2x2,3x3), never more than an order of magnitude, and through the whole tiny model it leaves every code index of the reference fixture
in place."""
import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.mark.parametrize("m", [2, 4])
def test_winograd_forms_equal_the_circular_conv_in_fp64(oracle, m):
    from oracle.winograd_sim import winograd_conv3x3_circular
    g = torch.Generator().manual_seed(5)
    for (B, C, O, H, W) in ((2, 8, 8, 8, 8), (1, 5, 7, 4, 12), (3, 16, 4, 16, 4)):
        x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(O, C, 3, 3, generator=g, dtype=torch.float64)
        ref = oracle.conv_circular3x3(x, w)
        got = winograd_conv3x3_circular(x, w, m)
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())


def test_fp32_rounding_of_the_winograd_forms(oracle):
    """128 channels on a 32 x 32 grid (the trunk's conv2): error against fp64 of the direct fp32 conv, F(2x2,3x3) and F(4x4,3x3)."""
    from oracle.winograd_sim import winograd_conv3x3_circular
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 128, 32, 32, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) / (9 * 128) ** 0.5
    exact = oracle.conv_circular3x3(x.double(), w.double())
    scale = float(exact.abs().max())
    err = {"direct": float((oracle.conv_circular3x3(x, w).double() - exact).abs().max()) / scale}
    for m in (2, 4):
        err[f"F({m},3)"] = float((winograd_conv3x3_circular(x, w, m).double() - exact).abs().max()) / scale
    print("max |err| / max |ref| of one fp32 conv2:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["direct"] <= 2e-6 and err["F(2,3)"] <= 2e-6
    assert err["F(4,3)"] <= 2e-5                        # a lone conv2 on unit-variance data: ~30x the direct form (1e-5 of the range);
    assert err["F(4,3)"] <= 60 * max(err["direct"], 1e-7)   # inside a Fixup block (residual + scaled branch) it is ~4x, see DESIGN.md


def test_f43_keeps_every_fixture_index_of_the_tiny_model(oracle, monkeypatch):
    """The whole tiny model with every 'same' block's conv2 as F(4x4,3x3) in fp32: indices of the reference fixture unchanged."""
    from oracle.winograd_sim import winograd_conv3x3_circular
    g = load_golden("model_tiny")
    spec = oracle.SPECS["tiny"]
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    x = oracle.make_patches(int(g["batch"]), 32, 0)
    direct = oracle.conv_circular3x3

    def conv(xx, w):
        if xx.shape[-1] % 4 == 0 and xx.shape[-2] % 4 == 0:
            return winograd_conv3x3_circular(xx, w, 4)
        return direct(xx, w)

    monkeypatch.setattr(oracle, "conv_circular3x3", conv)
    _, idx, _ = oracle.encoder_forward(x, p, spec)
    idx = idx[0] if isinstance(idx, (tuple, list)) else idx
    assert np.array_equal(np.asarray(idx).reshape(-1), g["idx"].astype(np.int64).reshape(-1))
