"""GPU parity of the 16-bit autocast path (BASELINE configs #3 bf16 / #4 fp16) against fixtures recorded
from the reference under torch.autocast('cpu', dtype) and against the CPU oracle under the same context.

Bar: the convolutions round operands and outputs to the 16-bit type with fp32 accumulation on both
sides, so a single kernel differs from the CPU only where an fp32 accumulation (different summation order)
lands within rounding noise of a 16-bit rounding boundary: isolated 1-ulp (2^-8 / 2^-11 relative) flips,
checked per block in test_autocast_blocks_match_oracle.  Through 136 residual blocks those flips (~1e-4
of the elements of every conv) act as noise of ~1e-3 relative on the pre-VQ activations, which is the
scale of the best/second-best margins of ~1-2 % of the rows: two correct bf16 evaluations with different
summation orders agree on ~98 % of indices, no more (the shallow models agree 100 %).  The end-to-end
tests therefore demand >= 96 % (bf16) / >= 99 % (fp16) agreement with the same-precision reference --
at least as close as that reference is to fp32 -- and report both numbers (bf16 flips
0.6-4 % of indices vs fp32, fp16 0.1-0.5 %)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TDT = {"bf16": torch.bfloat16, "f16": torch.float16}


def params_for(oracle, name, g):
    p = oracle.make_params(oracle.SPECS[name], 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    return p


@pytest.mark.parametrize("name,tag,size", [("tinyP", "bf16", 32), ("tiny", "f16", 32), ("B", "bf16", 256),
                                           ("A", "bf16", 512), ("C", "f16", 256)])
def test_autocast_forward_matches_reference_fixture(amd, oracle, name, tag, size):
    g = load_golden(f"model_{name}_{tag}")
    p = params_for(oracle, name, g)
    B = int(g["batch"])
    x = oracle.make_patches(B, size, 0)
    nat = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    out, idx, loss = nat.forward(x.cuda())
    torch.cuda.synchronize()
    ref = g["idx"].astype(np.int64)
    agree = float((idx.cpu().numpy() == ref).mean())
    agree32 = float((idx.cpu().numpy() == g["idx_fp32"].astype(np.int64)).mean())
    samp = out.cpu()[:, :, ::16, ::16]
    rel = float(((samp - torch.from_numpy(g["out_sample"])) ** 2).mean() / (torch.from_numpy(g["out_sample"]) ** 2).mean())
    print(f"cfg {name} {tag}: index agreement with the {tag} reference {agree * 100:.3f}% "
          f"(with the fp32 reference {agree32 * 100:.2f}%), relative output MSE {rel:.2e}, "
          f"loss {float(loss):.6f} vs {float(g['loss']):.6f}")
    assert agree >= (0.96 if tag == "bf16" else 0.99)
    assert agree >= agree32 - 0.005            # as close to the 16-bit reference as that one is to fp32
    assert rel <= (5e-2 if tag == "bf16" else 1e-2)
    assert abs(float(loss) - float(g["loss"])) <= 5e-3 * float(g["loss"])


@pytest.mark.parametrize("name,tag,size", [("A", "bf16", 512), ("C", "f16", 128), ("B", "bf16", 256)])
def test_autocast_features_no_further_from_exact_than_the_reference(amd, oracle, name, tag, size):
    """End-to-end pre-VQ features of the 68 / 74-block encoders at the fixture batch: HIP 16-bit, the reference's 16-bit
    evaluation (the oracle under CPU autocast: bit-identical to the reference, tests/test_oracle_golden.py) and an fp64
    evaluation with NO rounding points (fp32 weights widened) of the same input.  Criterion without a chosen fraction:
    ||hip16 - exact|| <= 1.25 ||ref16 - exact|| (RMS over all features, and max) -- the HIP path is no further from the
    true value than the reference's own 16-bit arithmetic is.  (cfg C: a 128 x 128 crop of the fixture patch -- the f16 autocast of
    the 256-channel model on the CPU, twice, is what bounds this test's run time; the encoder is fully convolutional.)"""
    from conftest import record_parity
    g = load_golden(f"model_{name}_{tag}")
    spec = oracle.SPECS[name]
    p = params_for(oracle, name, g)
    B = int(g["batch"])
    x = oracle.make_patches(B, max(size, 256), 0)[:, :, :size, :size].contiguous()
    nat = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    z_hip = nat.encode_features(x.cuda()).permute(0, 3, 1, 2).cpu().double()
    projected = spec.projection_dim > 0                 # the handle then reports the projected features (vq.py:190)
    vq = "encoder.vq_layers.0."
    with torch.autocast("cpu", dtype=TDT[tag]):
        z_ref = oracle.encoder_features(x, p, spec)
        if projected:
            z_ref = torch.nn.functional.conv2d(z_ref, p[vq + "proj_in.weight"], p[vq + "proj_in.bias"])
    z_ref = z_ref.double()
    p64 = {k: v.double() for k, v in p.items() if torch.is_tensor(v) and v.is_floating_point()}
    z_ex = oracle.encoder_features(x.double(), p64, spec)
    if projected:
        z_ex = torch.nn.functional.conv2d(z_ex, p64[vq + "proj_in.weight"], p64[vq + "proj_in.bias"])
    assert z_hip.shape == z_ex.shape, (z_hip.shape, z_ex.shape)
    eh, er = (z_hip - z_ex).abs(), (z_ref - z_ex).abs()
    rh, rr = float((eh ** 2).mean().sqrt()), float((er ** 2).mean().sqrt())
    mh, mr = float(eh.max()), float(er.max())
    scale = float(z_ex.abs().max())
    record_parity("features_vs_exact_fp64", model=name, dtype=tag, batch=B, rms_hip=rh / scale, rms_ref16=rr / scale,
                  max_hip=mh / scale, max_ref16=mr / scale, rms_ratio=rh / rr, max_ratio=mh / mr)
    assert rh <= 1.25 * rr and mh <= 1.25 * mr, (rh, rr, mh, mr)


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_autocast_blocks_match_oracle(amd, oracle, tag):
    """Per-kernel check on identical inputs: one Fixup block (fused and unfused paths) vs the oracle under
    CPU autocast.  Elementwise: equal up to isolated 1-ulp(16-bit) flips of conv outputs."""
    C, H, W, B = 32, 8, 32, 2
    g = torch.Generator().manual_seed(3)
    p = {}
    for n in ("bias1a", "bias1b", "bias2a", "bias2b", "bias3a", "bias3b", "bias4"):
        p["blk." + n] = (torch.rand(1, generator=g) - 0.5) * 0.4
    p["blk.scale"] = torch.rand(1, generator=g) + 0.5
    p["blk.branch_conv1.weight"] = torch.randn(C, C, 1, 1, generator=g) / C ** 0.5
    p["blk.branch_conv2.weight"] = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
    p["blk.branch_conv3.weight"] = torch.randn(C, C, 1, 1, generator=g) / C ** 0.5
    x = torch.randn(B, C, H, W, generator=g)
    with torch.autocast("cpu", dtype=TDT[tag]):
        ref = oracle.fixup_block(x, p, "blk", "same").float()
    xc = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed = [amd.ops.pack_conv_weight(p[f"blk.branch_conv{i}.weight"].cuda(), dtype=tag) for i in (1, 2, 3)]
    sc = [float(p["blk." + n]) for n in ("bias1a", "bias1b", "bias2a", "bias2b", "bias3a", "bias3b", "bias4", "scale")]
    fused = amd.ops.fixup_same_block(xc, *packed, sc, dtype=tag).permute(0, 3, 1, 2).cpu()
    L = amd._lib
    t = amd.ops.conv2d(xc, packed[0], C, 1, pre=(sc[0], sc[1]), act=(sc[2], sc[3]), dtype=tag)
    t = amd.ops.conv2d(t, packed[1], C, 3, 1, 1, L.PAD_CIRCULAR, act=(sc[4], sc[5]), dtype=tag)
    unf = amd.ops.conv2d(t, packed[2], C, 1, scale_bias=(sc[7], sc[6]), residual=xc, dtype=tag).permute(0, 3, 1, 2).cpu()
    ulp = 2.0 ** (-8 if tag == "bf16" else -11)
    for got in (fused, unf):
        err = (got - ref).abs()
        scale = float(ref.abs().max())
        assert float((err > 1e-5 * scale).float().mean()) <= 0.02          # few elements differ at all
        assert float(err.max()) <= 8 * ulp * scale                         # and only by 16-bit rounding flips


def test_module_mirror_honours_torch_autocast(amd, oracle):
    """`with torch.autocast('cuda', dtype=bf16): model.encoder(imgs)` (the reference's run_eval idiom)
    selects the bf16 autocast handle."""
    from vqae_amd.model import VQAE
    g = load_golden("model_tinyP_bf16")
    p = params_for(oracle, "tinyP", g)
    model = VQAE.from_spec(amd.SPECS["tinyP"])
    model.load_state_dict(p, strict=False)
    model = model.cuda().eval()
    x = oracle.make_patches(2, 32, 0).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        (q,), (idx,), (loss,) = model.encoder(x)
    (q32,), (idx32,), _ = model.encoder(x)
    assert float((idx.cpu().numpy() == g["idx"].astype(np.int64)).mean()) >= 0.99
    assert float((idx32.cpu().numpy() == g["idx_fp32"].astype(np.int64)).mean()) == 1.0
