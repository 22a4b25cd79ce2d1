"""GPU parity: conv-stack primitives (through the C ABI) against plain PyTorch fp32 on the CPU.
Tolerance: fp32 implicit GEMM vs oneDNN differ only in summation order -> |err| <= 2e-5 * scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def close(a, b, tol=2e-5):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


GEOMS = [  # (ksize, stride, pad, circular)
    (1, 1, 0, False), (3, 1, 1, True), (2, 2, 0, False), (3, 1, 1, False),
]


@pytest.mark.parametrize("ks,stride,pad,circ", GEOMS)
@pytest.mark.parametrize("cin,cout,hw,b", [(128, 128, 32, 3), (16, 16, 40, 2), (8, 16, 24, 2), (32, 64, 16, 5),
                                           (64, 32, 8, 3), (256, 256, 16, 1), (24, 40, 12, 2), (128, 8, 32, 1)])
def test_conv_geometries(amd, ks, stride, pad, circ, cin, cout, hw, b):
    L = amd._lib
    g = torch.Generator().manual_seed(cin * 31 + cout + ks)
    x = torch.randn(b, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    if circ:
        xp = F.pad(x, (1, 1, 1, 1), mode="circular")
        ref = F.conv2d(xp, w, stride=stride)
    else:
        ref = F.conv2d(x, w, stride=stride, padding=pad)
    wp = amd.ops.pack_conv_weight(w.cuda())
    y = amd.ops.conv2d(nhwc(x).cuda(), wp, cout, ks, stride, pad,
                       L.PAD_CIRCULAR if circ else (L.PAD_ZEROS if pad else L.PAD_NONE))
    torch.cuda.synchronize()
    close(nchw(y.cpu()), ref)


def test_conv_fixup_pre_and_epilogue(amd):
    """The fused Fixup ops: ELU(x+a)+b pre-op, *scale+bias+residual and ELU epilogues
    (conv_block.py:199-214), in the reference's rounding order."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 16, 16, generator=g)
    w = torch.randn(32, 32, 1, 1, generator=g) / 32 ** 0.5
    res = torch.randn(2, 32, 16, 16, generator=g)
    bv = torch.randn(32, generator=g)
    wp = amd.ops.pack_conv_weight(w.cuda())
    xc, rc = nhwc(x).cuda(), nhwc(res).cuda()
    # conv1-style
    ref = F.elu(F.conv2d(F.elu(x + 0.3) + (-0.2), w) + 0.11) + 0.07
    y = amd.ops.conv2d(xc, wp, 32, 1, pre=(0.3, -0.2), act=(0.11, 0.07))
    close(nchw(y.cpu()), ref)
    # conv3-style, in place over the residual
    ref = F.conv2d(x, w) * 1.3 + 0.05 + res
    y = amd.ops.conv2d(xc, wp, 32, 1, scale_bias=(1.3, 0.05), residual=rc, out=rc)
    close(nchw(y.cpu()), ref)
    # skip-style: conv(x + c) + d ; projection-style: per-channel bias
    ref = F.conv2d(x + 0.4, w) + 0.25
    close(nchw(amd.ops.conv2d(xc, wp, 32, 1, pre=(0.4,), bias_s=0.25).cpu()), ref)
    ref = F.conv2d(x, w, bv)
    close(nchw(amd.ops.conv2d(xc, wp, 32, 1, bias_vec=bv.cuda()).cpu()), ref)


def test_conv_ragged_m_and_empty(amd):
    g = torch.Generator().manual_seed(4)
    w = torch.randn(16, 8, 3, 3, generator=g) / 72 ** 0.5
    wp = amd.ops.pack_conv_weight(w.cuda())
    for b, hw in [(1, 5), (3, 7), (1, 11)]:                      # M = 25, 147, 121: not multiples of 128
        x = torch.randn(b, 8, hw, hw, generator=g)
        ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="circular"), w)
        y = amd.ops.conv2d(nhwc(x).cuda(), wp, 16, 3, 1, 1, amd._lib.PAD_CIRCULAR)
        close(nchw(y.cpu()), ref)
    y = amd.ops.conv2d(torch.zeros(0, 4, 4, 8).cuda(), wp, 16, 3, 1, 1, amd._lib.PAD_CIRCULAR)
    assert y.shape == (0, 4, 4, 16)


def test_mfma_operand_layout_asymmetric(amd):
    """A = identity-like input with an asymmetric weight catches swapped C/D row<->col maps."""
    cin = cout = 64
    x = torch.zeros(1, cin, 8, 16)
    for c in range(cin):
        x[0, c, c % 8, (c * 3) % 16] = 1.0 + c
    w = (torch.arange(cout * cin, dtype=torch.float32).reshape(cout, cin, 1, 1) % 97) / 97.0
    ref = F.conv2d(x, w)
    y = amd.ops.conv2d(nhwc(x).cuda(), amd.ops.pack_conv_weight(w.cuda()), cout, 1)
    close(nchw(y.cpu()), ref, 1e-6)


@pytest.mark.parametrize("cin,cout", [(3, 8), (3, 16), (3, 32), (8, 3), (16, 3), (32, 3)])
def test_stem_conv_direct(amd, cin, cout):
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, 20, 28, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    y = amd.ops.conv3x3_direct(nhwc(x).cuda(), w.cuda(), b.cuda())
    close(nchw(y.cpu()), F.conv2d(x, w, b, padding=1))


def test_stem_conv_uint8_ingest(amd, oracle):
    u8 = oracle.make_patches_u8(2, 32, 3)
    x = oracle.normalize_u8(u8)
    g = torch.Generator().manual_seed(0)
    w = torch.randn(16, 3, 3, 3, generator=g) / 27 ** 0.5
    b = torch.randn(16, generator=g)
    mean = [m * 255 for m in oracle.MEAN]
    inv = [1.0 / (s * 255) for s in oracle.STD]
    y = amd.ops.conv3x3_direct(None, w.cuda(), b.cuda(), x_u8=torch.from_numpy(u8).cuda(), mean255=mean,
                               inv_std255=inv)
    close(nchw(y.cpu()), F.conv2d(x, w, b, padding=1))


@pytest.mark.parametrize("shape", [(2, 8, 9, 7), (1, 32, 16, 16), (3, 4, 1, 5)])
def test_bicubic_up2(amd, oracle, shape):
    x = torch.randn(*shape)
    y = nchw(amd.ops.bicubic_up2(nhwc(x).cuda(), 0.125).cpu())
    ref = oracle.bicubic_up2_explicit(x + 0.125)
    assert torch.equal(y, ref), float((y - ref).abs().max())         # same taps, same order: bit-exact
    close(y, oracle.bicubic_up2(x + 0.125), 1e-6)                     # ATen's own kernel: <= 1 ulp-ish


@pytest.mark.parametrize("cin,cout,bias,shape", [(32, 16, False, (2, 12, 20)), (8, 8, True, (1, 5, 7)), (64, 32, False, (3, 16, 16))])
def test_resize_conv2d_mirror_matches_upsample_then_conv(amd, cin, cout, bias, shape):
    """The stand-alone mirror vqae_amd.layers.conv.ResizeConv2D (reference vq_ae/layers/conv.py:4-11: nn.Conv2d whose forward
    is conv(nn.Upsample(scale_factor=2, mode='bicubic')(x))) against exactly that composition in PyTorch fp32 on the CPU."""
    from vqae_amd.layers.conv import ResizeConv2D
    torch.manual_seed(cin + cout)
    m = ResizeConv2D(cin, cout, kernel_size=1, bias=bias)
    ref_conv = torch.nn.Conv2d(cin, cout, 1, bias=bias)
    ref_conv.load_state_dict(m.state_dict())                           # same parameter names as nn.Conv2d: weight (, bias)
    B, H, W = shape
    x = torch.randn(B, cin, H, W)
    with torch.no_grad():
        want = ref_conv(torch.nn.Upsample(scale_factor=2, mode="bicubic")(x))
        got = m.cuda()(x.cuda()).cpu()
    assert got.shape == want.shape == (B, cout, 2 * H, 2 * W)
    close(got, want, 2e-5)
    with pytest.raises(NotImplementedError):
        ResizeConv2D(cin, cout, kernel_size=3, padding=1)


def test_layout_roundtrip_and_helpers(amd, oracle):
    x = torch.randn(3, 20, 6, 10)
    y = amd.ops.nchw_to_nhwc(x.cuda())
    assert torch.equal(y.cpu(), nhwc(x))
    assert torch.equal(amd.ops.nhwc_to_nchw(y).cpu(), x)
    lab = (torch.rand(3, 64, 64) > 0.97).to(torch.uint8)
    pooled = amd.ops.label_maxpool(lab.cuda(), 32).cpu().numpy()
    assert np.array_equal(pooled, oracle.adaptive_max_pool_labels(lab.numpy(), 32))


def test_stitch_tiles_matches_driver_fixture(amd, oracle):
    from conftest import load_golden
    g = load_golden("driver")
    tiles, meta, sizes = g["tiles"], g["meta"], g["sizes"]
    for s, name in enumerate(["images/slide_a", "images/slide_b"]):
        sel = meta[:, 0] == s
        r, c = sizes[s]
        grid = torch.full((int(r) * 4, int(c) * 4), -1, dtype=torch.int64).cuda()
        amd.ops.stitch_tiles(torch.from_numpy(tiles[sel]).cuda(), torch.from_numpy(meta[sel][:, 1:]).cuda(), grid)
        out = oracle.cast_to_lowest_dtype(grid.cpu().numpy())
        assert out.dtype == g["grid:" + name].dtype and np.array_equal(out, g["grid:" + name])


@pytest.mark.parametrize("C,H,W,B", [(16, 16, 32, 2), (16, 8, 64, 3), (32, 8, 32, 2), (32, 12, 64, 1), (16, 256, 256, 1), (8, 16, 64, 2), (8, 512, 512, 1)])
def test_fused_same_block_matches_oracle(amd, oracle, C, H, W, B):
    """Fused 1x1 -> 3x3 circular -> 1x1 Fixup block (one launch) vs the CPU oracle's fixup_block."""
    assert amd.ops.fixup_same_supported(C, H, W)
    g = torch.Generator().manual_seed(C + H + W)
    p = {}
    for n in ("bias1a", "bias1b", "bias2a", "bias2b", "bias3a", "bias3b", "bias4"):
        p["blk." + n] = (torch.rand(1, generator=g) - 0.5) * 0.4
    p["blk.scale"] = torch.rand(1, generator=g) + 0.5
    p["blk.branch_conv1.weight"] = torch.randn(C, C, 1, 1, generator=g) / C ** 0.5
    p["blk.branch_conv2.weight"] = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
    p["blk.branch_conv3.weight"] = torch.randn(C, C, 1, 1, generator=g) / C ** 0.5
    x = torch.randn(B, C, H, W, generator=g)
    ref = oracle.fixup_block(x, p, "blk", "same")
    packed = [amd.ops.pack_conv_weight(p[f"blk.branch_conv{i}.weight"].cuda()) for i in (1, 2, 3)]
    sc = [float(p["blk." + n]) for n in ("bias1a", "bias1b", "bias2a", "bias2b", "bias3a", "bias3b", "bias4", "scale")]
    y = amd.ops.fixup_same_block(nhwc(x).cuda(), *packed, sc)
    torch.cuda.synchronize()
    close(nchw(y.cpu()), ref)


def test_fused_same_block_unsupported_shapes(amd):
    assert not amd.ops.fixup_same_supported(128, 32, 32)       # trunk: stays on the 3-launch MFMA path
    assert not amd.ops.fixup_same_supported(16, 8, 16)         # W % 32 != 0
    with pytest.raises(NotImplementedError):
        amd.ops.fixup_same_block(torch.zeros(1, 8, 16, 16).cuda(), torch.zeros(8).cuda(), torch.zeros(8).cuda(),
                                 torch.zeros(8).cuda(), [0] * 8)


@pytest.mark.parametrize("tag,dt", [("bf16", torch.bfloat16), ("f16", torch.float16)])
@pytest.mark.parametrize("c0,H,W", [(8, 16, 128), (16, 8, 64), (32, 24, 192)])
def test_stem16_mfma_stems_match_autocast_reference(amd, oracle, tag, dt, c0, H, W):
    """The 16-bit-MFMA stems (csrc/stem16.hip; taken for 16-bit dtypes when H % 8 == 0 and W % 64 == 0) against torch's conv
    under CPU autocast -- operands and bias rounded to the 16-bit type, fp32 accumulation, rounded result -- on identical
    inputs: in-stem from fp32 NHWC and from uint8 + normalisation, out-stem to 3 channels.  Zero padding at every border
    (the tiles touch all four).  Equal up to isolated 1-ulp(16) flips of the result (summation order)."""
    ulp = 2.0 ** -8 if tag == "bf16" else 2.0 ** -11
    g = torch.Generator().manual_seed(c0 + H)

    def check(y, ref):
        ref = ref.float()
        err = (nchw(y.cpu()) - ref).abs()
        scale = max(1.0, float(ref.abs().max()))
        assert float(err.max()) <= 1.01 * ulp * scale, float(err.max()) / scale
        assert float((err > 1e-6 * scale).float().mean()) <= 2e-3

    # in-stem, fp32 NHWC input
    x = torch.randn(3, 3, H, W, generator=g)
    w = torch.randn(c0, 3, 3, 3, generator=g) / 27 ** 0.5
    b = torch.randn(c0, generator=g)
    with torch.autocast("cpu", dtype=dt):
        ref = F.conv2d(x, w, b, padding=1)
    check(amd.ops.conv3x3_direct(nhwc(x).cuda(), w.cuda(), b.cuda(), dtype=tag), ref)
    # in-stem, uint8 input normalised on the device
    u8 = oracle.make_patches_u8(2, max(H, W), 5)[:, :H, :W].copy()
    xn = oracle.normalize_u8(u8)
    mean = [m * 255 for m in oracle.MEAN]
    inv = [1.0 / (s * 255) for s in oracle.STD]
    with torch.autocast("cpu", dtype=dt):
        ref = F.conv2d(xn, w, b, padding=1)
    y = amd.ops.conv3x3_direct(None, w.cuda(), b.cuda(), x_u8=torch.from_numpy(u8).cuda(), mean255=mean, inv_std255=inv, dtype=tag)
    ref = ref.float()
    err = (nchw(y.cpu()) - ref).abs()
    scale = max(1.0, float(ref.abs().max()))
    assert float(err.max()) <= 2.01 * ulp * scale      # the normalisation's own fp32 rounding can move an operand by one ulp(16)
    assert float((err > 1e-6 * scale).float().mean()) <= 2e-2
    # out-stem
    xo = torch.randn(2, c0, H, W, generator=g)
    wo = torch.randn(3, c0, 3, 3, generator=g) / (9 * c0) ** 0.5
    bo = torch.randn(3, generator=g)
    with torch.autocast("cpu", dtype=dt):
        ref = F.conv2d(xo, wo, bo, padding=1)
    check(amd.ops.conv3x3_direct(nhwc(xo).cuda(), wo.cuda(), bo.cuda(), dtype=tag), ref)


@pytest.mark.parametrize("mode,cin,cout,shape", [("same", 16, 32, (2, 8, 32)), ("same", 32, 16, (1, 12, 20)), ("out", 16, 8, (2, 8, 32)),
                                                 ("out", 16, 16, (2, 8, 32)), ("out", 24, 40, (1, 6, 10))])
def test_fixup_block_shape_variants_match_oracle(amd, oracle, mode, cin, cout, shape):
    """Block shapes the reference class accepts beyond what Encoder / Decoder compose (conv_block.py:180-191 with
    pre_activation_fixup.yaml): 'same' with in_channels != out_channels (1x1 skip_conv on inp + bias1c, + bias1d) and mode 'out'
    (3x3 circular branch_conv2; 3x3 ZERO-padded skip_conv, or the identity when the channel counts agree): the mirror module
    against the oracle's restatement of PreActFixupResBlock.forward (conv_block.py:196-216), fp32."""
    from vqae_amd.layers.conv_block import PreActFixupResBlock
    torch.manual_seed(cin * 100 + cout)
    m = PreActFixupResBlock(cin, cout, mode, bottleneck_divisor=1, n_layers=4)
    with torch.no_grad():
        for n, prm in m.named_parameters():
            if n.endswith("branch_conv3.weight"):
                prm.normal_(0.0, 1.0 / prm.shape[1] ** 0.5)                # the reference zero-initialises conv3: make the branch count
            elif prm.numel() == 1:
                prm.copy_(torch.rand(1) * 0.4 - 0.2 + (1.0 if n == "scale" else 0.0))
    assert (m.skip_conv is None) == (cin == cout)
    if m.skip_conv is not None:
        assert m.skip_conv.weight.shape[-1] == (1 if mode == "same" else 3)
    p = {"blk." + k: v.detach().clone() for k, v in m.state_dict().items()}
    B, H, W = shape
    x = torch.randn(B, cin, H, W)
    want = oracle.fixup_block(x, p, "blk", mode)
    got = m.cuda()(x.cuda()).cpu()
    assert got.shape == want.shape == (B, cout, H, W)
    close(got, want, 2e-5)
