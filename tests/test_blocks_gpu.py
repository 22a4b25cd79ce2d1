"""GPU parity of the PRODUCTION block kernels, block by block, on identical inputs.

`vqae_run_blocks` (include/vqae_hip.h) runs blocks [first, first + count) of a handle's encoder / decoder list
through exactly the dispatch the whole-model calls use -- wino_trunk_kernel<128|64|32>, fixup_conv1_kernel,
down_block_kernel<16|32|64>, up_tail_kernel, the fused small-channel kernels, the 16-bit trunk kernels, the fused
conv2+conv3+next-conv1 tails when count > 1 -- i.e. the kernels `test_blocks_match_reference_taps` (module mirrors ->
generic kernels) never reaches.

Oracle link: tests/golden/taps_<model>_<dtype>.npz hold, for every block of three mid-size models, a strided sample
and the fp64 checksums of the REFERENCE's output (fp32 and under torch.autocast bf16 / f16,
scripts/extract_embeddings/extract_embeddings.py:124-125); the CPU suite proves the oracle reproduces them bit for
bit (tests/test_oracle_golden.py::test_block_taps_match_reference_fixture).  Here the oracle recomputes the full
tensors on the box's CPU, they are re-checked against the fixture samples, and each HIP block is fed the oracle's
block input and compared with the oracle's block output over the FULL tensor.

Principled 16-bit bar (round 3): next to the ulp16 bars every 16-bit block is also evaluated in fp64 WITHOUT the autocast
rounding points (fp32 weights widened, no operand / output rounding) on the identical input -- the value both 16-bit
evaluations approximate -- and the HIP block may be no further from it than 1.25x the reference's own 16-bit evaluation
(RMS and max): no hand-picked fraction enters that criterion.

Bars: fp32 -- max |err| <= 2e-5 * max(1, max|ref|) per block (summation order differs from oneDNN's);
16-bit -- conv operands / outputs are rounded to the 16-bit type on both sides, so outputs are equal except where an
fp32 accumulation lands within summation-order noise of a 16-bit rounding boundary: isolated 1-ulp(16) flips of a
conv output.  A block has three chained convs and one flipped conv2 input moves 9 C products by a fraction of an
ulp16 each, so a flip in conv1 breeds a few more downstream (measured: up to 4.9 % of a block's elements at C = 256 in
f16, 0.7 % in bf16): <= 8 % of the elements may differ by more than 1e-5 * scale, and NONE by more than 3 ulp16 * scale
(measured worst: 1.3 ulp16; a wrong weight, lane or pixel gives errors of the order of the branch itself, 30-300 ulp16).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity, tap_sample

pytestmark = pytest.mark.gpu
TDT = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}
ULP = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}
_cache = {}


def oracle_taps(oracle, name, tag):
    """Full per-block tensors of the oracle on this machine's CPU, verified against the reference's fixture."""
    key = (name, tag)
    if key in _cache:
        return _cache[key]
    g = load_golden(f"taps_{name}_{tag}")
    spec = oracle.SPECS[name]
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    x = oracle.make_patches(int(g["batch"]), int(g["size"]), 0)
    # encoder taps from x; decoder taps from the REFERENCE's indices (q = codebook lookup [+ proj_out]): on another CPU
    # the 16-bit encoder flips a few near-tie indices, which would change q wholesale and make the decoder taps
    # incomparable with the fixture.  (embed[idx] differs from the reference's x + (q - x) by <= 1 fp32 ulp.)
    import contextlib
    taps = {}
    with (torch.autocast("cpu", dtype=TDT[tag]) if TDT[tag] is not None else contextlib.nullcontext()):
        oracle.encoder_forward(x, p, spec, taps)
        vq = "encoder.vq_layers.0."
        q = p[vq + "embed"][torch.from_numpy(g["idx"].astype(np.int64))].permute(0, 3, 1, 2).contiguous()
        if spec.projection_dim > 0:
            q = torch.nn.functional.conv2d(q, p[vq + "proj_out.weight"], p[vq + "proj_out.bias"])
        taps["q"] = q
        taps["out"] = oracle.decoder_forward((q,), p, spec, taps)
    # the GPU box's CPU may take other oneDNN kernels than the container the fixture was recorded in: allow
    # summation-order noise (fp32) / isolated 16-bit rounding flips here; bit-exactness is the CPU suite's job
    for k in g.files:
        if not k.startswith("tap:"):
            continue
        got, ref = tap_sample(taps[k[4:]].float()).numpy(), g[k]
        scale = max(1.0, float(np.abs(ref).max()))
        err = np.abs(got - ref)
        if tag == "f32":
            assert err.max() <= 1e-5 * scale, (k, err.max())
        else:                # the taps are cumulative (each block runs on the chain's own previous output): flips add up
            rms = float(np.sqrt((err.astype(np.float64) ** 2).mean() / (ref.astype(np.float64) ** 2).mean()))
            assert rms <= 4 * ULP[tag], (k, rms)
    _cache[key] = (spec, p, x, {k: v.float() for k, v in taps.items() if torch.is_tensor(v)})
    return _cache[key]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def compare(got_nhwc, ref_nchw, tag, n_blocks):
    got = got_nhwc.permute(0, 3, 1, 2).cpu()
    err = (got - ref_nchw).abs()
    scale = max(1.0, float(ref_nchw.abs().max()))
    mx = float(err.max())
    frac = float((err > 1e-5 * scale).float().mean())
    if tag == "f32":
        ok = mx <= 2e-5 * scale * n_blocks
    else:
        ok = frac <= 0.08 * n_blocks and mx <= 3 * ULP[tag] * scale * n_blocks
    return ok, mx / scale, frac


_p64 = {}


def exact_block(oracle, p, name, x_nchw, prefix, mode, spec):
    """The block in fp64 with no rounding points: what the 16-bit evaluations (reference autocast and HIP) approximate."""
    if name not in _p64:
        _p64.clear()                                   # one model's fp64 weights at a time
        _p64[name] = {k: v.double() for k, v in p.items() if torch.is_tensor(v) and v.is_floating_point()}
    return oracle.conv_block(x_nchw.double(), _p64[name], prefix, mode, spec)


def distance_to_exact(got_nhwc, ref_nchw, exact_nchw):
    """(rms_hip, rms_ref, max_hip, max_ref) of |. - exact| in fp64."""
    got = got_nhwc.permute(0, 3, 1, 2).cpu().double()
    eh, er = (got - exact_nchw).abs(), (ref_nchw.double() - exact_nchw).abs()
    return float((eh ** 2).mean().sqrt()), float((er ** 2).mean().sqrt()), float(eh.max()), float(er.max())


def block_lists(oracle, spec):
    return (("encoder", oracle.encoder_blocks(spec), "stem"), ("decoder", oracle.decoder_blocks(spec), "q"))


@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["mid", "mid16", "midA", "midC", "midW"])
def test_production_blocks_match_oracle_per_block(amd, oracle, name, tag):
    """Every block alone (count = 1): conv1 launch + fused tail without the next-block conv1, 'down' / 'up' blocks."""
    spec, p, x, taps = oracle_taps(oracle, name, tag)
    nat = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=None if tag == "f32" else tag)
    worst = (0.0, 0.0, "")
    bad, far = [], []
    worst_ratio = (0.0, 0.0, "")                        # (rms ratio, max ratio, block) of |hip - exact| / |ref16 - exact|
    for side, blocks, first_in in block_lists(oracle, spec):
        assert nat.block_count(side) == len(blocks)
        prev = taps[first_in]
        for i, (prefix, mode, ci, co) in enumerate(blocks):
            y = nat.run_blocks(side, i, 1, nhwc(prev).cuda())
            ok, rel, frac = compare(y, taps[prefix], tag, 1)
            if not ok:
                bad.append((prefix, mode, ci, co, rel, frac))
            if rel > worst[0]:
                worst = (rel, frac, prefix)
            if tag != "f32":
                ex = exact_block(oracle, p, name, prev, prefix, mode, spec)
                rh, rr, mh, mr = distance_to_exact(y, taps[prefix], ex)
                ratio = (rh / rr, mh / mr)
                if ratio[0] > worst_ratio[0]:
                    worst_ratio = (ratio[0], ratio[1], prefix)
                if rh > 1.25 * rr or mh > 1.25 * mr:
                    far.append((prefix, mode, ci, co, rh, rr, mh, mr))
            prev = taps[prefix]
    record_parity("blocks_per_block", model=name, dtype=tag, blocks=sum(len(b) for _, b, _ in block_lists(oracle, spec)),
                  failed=len(bad), worst_rel_err=worst[0], worst_frac_off=worst[1], worst_block=worst[2],
                  **({} if tag == "f32" else dict(further_from_fp64_than_ref=len(far), worst_rms_ratio_vs_ref16=worst_ratio[0],
                                                  its_max_ratio=worst_ratio[1], worst_ratio_block=worst_ratio[2])))
    assert not bad, bad
    assert not far, far                                # the HIP block is as close to the exact value as the reference's 16-bit one


def autocast_ctx(tag):
    import contextlib
    return torch.autocast("cpu", dtype=TDT[tag]) if TDT[tag] is not None else contextlib.nullcontext()


@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["mid", "mid16", "midA", "midC", "midW"])
def test_fused_next_conv1_on_identical_inputs(amd, oracle, name, tag):
    """The cross-block fusion (conv2 + conv3 + the NEXT block's conv1 in one launch; t1 handed from launch to launch,
    as 16-bit in the 16-bit modes) on identical inputs: for every pair of consecutive blocks, y1 = blocks[i] alone and
    y2 = blocks[i, i+1] in one run (block i+1 then takes its t1 from block i's fused tail); the oracle evaluates block
    i+1 on the GPU's own y1, so y2 is compared with a reference that saw bit-identical input.  Per-block bars."""
    spec, p, x, taps = oracle_taps(oracle, name, tag)
    nat = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=None if tag == "f32" else tag)
    bad, worst, n = [], (0.0, 0.0, ""), 0
    for side, blocks, first_in in block_lists(oracle, spec):
        ins = [taps[first_in]] + [taps[b[0]] for b in blocks[:-1]]
        for i in range(len(blocks) - 1):
            xin = nhwc(ins[i]).cuda()
            y1 = nat.run_blocks(side, i, 1, xin)
            y2 = nat.run_blocks(side, i, 2, xin)
            with autocast_ctx(tag):
                ref2 = oracle.conv_block(y1.permute(0, 3, 1, 2).cpu(), p, blocks[i + 1][0], blocks[i + 1][1], spec).float()
            ok, rel, frac = compare(y2, ref2, tag, 1)
            n += 1
            if not ok:
                bad.append((blocks[i + 1][0], blocks[i + 1][1], rel, frac))
            if rel > worst[0]:
                worst = (rel, frac, blocks[i + 1][0])
    record_parity("fused_pairs_identical_input", model=name, dtype=tag, pairs=n, failed=len(bad), worst_rel_err=worst[0],
                  worst_frac_off=worst[1], worst_block=worst[2])
    assert not bad, bad


@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["mid", "mid16", "midA", "midC", "midW"])
def test_production_block_chains_match_oracle(amd, oracle, name, tag):
    """Every maximal chain of same-width 'same' blocks in one run (the production dispatch: chain-head conv1 + one
    fused launch per block) against the oracle's chain.  In fp32 the per-block bar scales with the chain length.  In
    16-bit a chain is NOT comparable element by element: one rounding flip at a conv input moves ~9 C products by a
    fraction of an ulp16 each and so flips further outputs (a branching process with mean > 1), i.e. after a few convs
    most elements differ -- by about one ulp16 of the branch.  The bar there is on magnitude: no element further than
    4 ulp16 * sqrt(blocks) * scale, mean |err| <= ulp16 * scale / 2 (a wrong weight / lane gives errors of the order
    of the branch itself, 10-100x more)."""
    spec, p, x, taps = oracle_taps(oracle, name, tag)
    nat = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=None if tag == "f32" else tag)
    bad, worst, n_runs = [], (0.0, 0.0, ""), 0
    for side, blocks, first_in in block_lists(oracle, spec):
        ins = [taps[first_in]] + [taps[b[0]] for b in blocks[:-1]]
        runs, i = [], 0
        while i < len(blocks):                         # maximal chains of same-width 'same' blocks
            j = i
            while j + 1 < len(blocks) and blocks[j + 1][1] == "same" and blocks[j][1] == "same" and blocks[j + 1][2] == blocks[i][2]:
                j += 1
            if j - i + 1 >= 2:
                runs.append((i, j - i + 1))
            i = j + 1
        for first, count in runs:
            y = nat.run_blocks(side, first, count, nhwc(ins[first]).cuda()).permute(0, 3, 1, 2).cpu()
            ref = taps[blocks[first + count - 1][0]]
            err = (y - ref).abs()
            scale = max(1.0, float(ref.abs().max()))
            mx, mean = float(err.max()) / scale, float(err.mean()) / scale
            ok = mx <= 2e-5 * count if tag == "f32" else (mx <= 4 * ULP[tag] * count ** 0.5 and mean <= 0.5 * ULP[tag])
            n_runs += 1
            if not ok:
                bad.append((blocks[first][0], count, mx, mean))
            if mx > worst[0]:
                worst = (mx, mean, f"{blocks[first][0]}+{count}")
    record_parity("block_chains", model=name, dtype=tag, runs=n_runs, failed=len(bad), worst_max_rel_err=worst[0],
                  its_mean_rel_err=worst[1], worst_run=worst[2])
    assert not bad, bad


@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["mid", "mid16", "midA", "midC", "midW"])
def test_mid_models_end_to_end_vs_reference(amd, oracle, name, tag):
    """Whole forward of the mid-size models against the reference's recorded indices / output samples."""
    g = load_golden(f"taps_{name}_{tag}")
    spec, p, x, taps = oracle_taps(oracle, name, tag)
    nat = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=None if tag == "f32" else tag)
    out, idx, loss = nat.forward(x.cuda())
    ref_idx = g["idx"].astype(np.int64)
    agree = float((idx.cpu().numpy() == ref_idx).mean())
    # reconstruction check that never depends on index agreement: decode the REFERENCE's indices
    dec = nat.decode_indices(torch.from_numpy(ref_idx).cuda()).cpu()
    ref_out = torch.from_numpy(g["tap:out"])
    rel = float(((tap_sample(dec) - ref_out) ** 2).mean() / (ref_out ** 2).mean())
    record_parity("mid_end_to_end", model=name, dtype=tag, idx_agreement=agree, n=int(ref_idx.size),
                  decode_ref_idx_rel_mse=rel, loss=float(loss), loss_ref=float(g["loss"]))
    if tag == "f32":
        assert agree >= 0.999 and rel <= 1e-9
    else:
        assert agree >= 0.97 and rel <= (2e-3 if tag == "bf16" else 1e-4)


@pytest.mark.parametrize("tag", ["bf16", "f16"])
@pytest.mark.parametrize("name", ["mid", "mid16", "midA"])
def test_down16_equals_fp32_mfma_form_on_identical_inputs(amd, oracle, name, tag, monkeypatch):
    """The 16-bit-MFMA 'down' block kernel (down16.hip) against the fp32-MFMA form with compiled-in cast points
    (down_fused.hip, VQAE_NO_DOWN16=1) on identical inputs: both multiply the same 16-bit-rounded operands exactly and
    accumulate in fp32, so they differ only where the summation order moves an fp32 sum across a 16-bit rounding
    boundary (isolated flips): per-block bars."""
    spec, p, x, taps = oracle_taps(oracle, name, tag)
    nat16 = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    monkeypatch.setenv("VQAE_NO_DOWN16", "1")
    nat32 = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    monkeypatch.delenv("VQAE_NO_DOWN16")
    blocks = oracle.encoder_blocks(spec)
    ins = [taps["stem"]] + [taps[b[0]] for b in blocks[:-1]]
    n = 0
    for i, (prefix, mode, ci, co) in enumerate(blocks):
        if mode != "down":
            continue
        xin = nhwc(ins[i]).cuda()
        a, b = nat16.run_blocks("encoder", i, 1, xin), nat32.run_blocks("encoder", i, 1, xin)
        ok, rel, frac = compare(a, b.permute(0, 3, 1, 2).cpu(), tag, 1)
        record_parity("down16_vs_fp32_mfma", model=name, dtype=tag, block=prefix, cin=ci, rel_err=rel, frac_off=frac)
        assert ok, (prefix, ci, rel, frac)
        assert not torch.equal(a, torch.zeros_like(a))
        n += 1
    assert n >= 2


@pytest.mark.parametrize("tag", ["bf16", "f16"])
@pytest.mark.parametrize("name", ["mid", "mid16", "midA"])
def test_up16_equals_unfused_form_on_identical_inputs(amd, oracle, name, tag, monkeypatch):
    """The fused 16-bit 'up' block (head16 conv1 + up16.hip: both resizes, conv2, conv3, skip_conv in one launch) against
    the unfused form (VQAE_NO_UP16=1: fp32 bicubic launches + generic convs with cast points) on identical inputs.  Same
    rounding points and the same resize arithmetic; only the convs' summation order differs: per-block bars."""
    spec, p, x, taps = oracle_taps(oracle, name, tag)
    fused = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    monkeypatch.setenv("VQAE_NO_UP16", "1")
    plain = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    monkeypatch.delenv("VQAE_NO_UP16")
    blocks = oracle.decoder_blocks(spec)
    ins = [taps["q"]] + [taps[b[0]] for b in blocks[:-1]]
    n = 0
    for i, (prefix, mode, ci, co) in enumerate(blocks):
        if mode != "up":
            continue
        xin = nhwc(ins[i]).cuda()
        a, b = fused.run_blocks("decoder", i, 1, xin), plain.run_blocks("decoder", i, 1, xin)
        ok, rel, frac = compare(a, b.permute(0, 3, 1, 2).cpu(), tag, 1)
        record_parity("up16_vs_unfused", model=name, dtype=tag, block=prefix, cin=ci, rel_err=rel, frac_off=frac)
        assert ok, (prefix, ci, rel, frac)
        n += 1
    assert n >= 2


@pytest.mark.parametrize("B,H,W", [(3, 128, 192), (1, 64, 256), (5, 192, 64)])
@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_16bit_down_up_kernels_on_odd_batches_and_non_square_inputs(amd, oracle, tag, B, H, W, monkeypatch):
    """down16 / up16 / 512-pixel trunk16 tiles on shapes the fixtures do not have (odd batches, H != W, levels whose width is
    not a multiple of the fused kernels' tile and therefore fall back to the generic launches in the middle of the model):
    the default handle against one with the block fusions switched off, on the same input.  Encoder features and decoder
    output: 16-bit chain bars (both are the same rounding-point arithmetic; flips cascade along the chain)."""
    name = "mid16"
    spec = oracle.SPECS[name]
    p = oracle.make_params(spec, 0)
    fused = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    for k in ("VQAE_NO_DOWN16", "VQAE_NO_UP16", "VQAE_NO_DOWN_FUSION"):
        monkeypatch.setenv(k, "1")
    plain = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=tag)
    for k in ("VQAE_NO_DOWN16", "VQAE_NO_UP16", "VQAE_NO_DOWN_FUSION"):
        monkeypatch.delenv(k)
    x = oracle.make_patches(B, 256, 21)[:, :, :H, :W].contiguous().cuda()
    z_f, z_p = fused.encode_features(x), plain.encode_features(x)
    assert z_f.shape == z_p.shape and bool(torch.isfinite(z_f).all())
    scale = float(z_p.abs().max())
    e = (z_f - z_p).abs()
    q = plain.encode(x)[0]
    d_f, d_p = fused.decode(q), plain.decode(q)
    ed = (d_f - d_p).abs()
    dscale = float(d_p.abs().max())
    record_parity("odd_geometry_16bit", dtype=tag, B=B, H=H, W=W, z_max_rel=float(e.max()) / scale, z_mean_rel=float(e.mean()) / scale,
                  d_max_rel=float(ed.max()) / dscale, d_mean_rel=float(ed.mean()) / dscale)
    n_blocks = 12
    assert float(e.max()) <= 4 * ULP[tag] * scale * np.sqrt(n_blocks) and float(e.mean()) <= 0.5 * ULP[tag] * scale
    assert float(ed.max()) <= 4 * ULP[tag] * dscale * np.sqrt(n_blocks) and float(ed.mean()) <= 0.5 * ULP[tag] * dscale


@pytest.mark.parametrize("name", ["mid", "midC", "midW"])
def test_persistent_conv1_equals_one_shot_bitwise(amd, oracle, name, monkeypatch):
    """fixup_conv1p_kernel (persistent workgroups, next tile's rows prefetched under the MFMAs; taken from 4 tiles per CU
    on) against the one-shot kernel on the fp32 chain heads of the mid models: same MFMA order -> bit-identical."""
    spec, p, x, taps = oracle_taps(oracle, name, "f32")
    nat = amd.NativeVQAE(amd.SPECS[name], p)
    n = 0
    for side, blocks, first_in in block_lists(oracle, spec):
        ins = [taps[first_in]] + [taps[b[0]] for b in blocks[:-1]]
        for i, (prefix, mode, ci, co) in enumerate(blocks):
            if mode != "same" or ci not in (32, 64, 128):
                continue
            xin = nhwc(ins[i]).cuda()
            monkeypatch.setenv("VQAE_CONV1_ONESHOT", "1")
            a = nat.run_blocks(side, i, 1, xin)
            monkeypatch.delenv("VQAE_CONV1_ONESHOT")
            monkeypatch.setenv("VQAE_CONV1_PERSIST_MIN_TILES", "1")
            b = nat.run_blocks(side, i, 1, xin)
            monkeypatch.delenv("VQAE_CONV1_PERSIST_MIN_TILES")
            assert torch.equal(a, b), (prefix, ci)
            n += 1
    assert n >= 2


def test_wino16_block_equals_direct_form(amd, oracle, monkeypatch):
    """fixup_same_wino16_kernel (C = 16, fp32: conv2 as Winograd F(2x2, 3x3), everything after conv1 in registers) against the
    direct 9-tap kernel (VQAE_NO_WINO16=1) on identical inputs, incl. several tiles per image in both directions and a batch
    that does not fill the persistent grid: fp32 rounding only (<= 2e-5 * scale; measured ~1e-6)."""
    g = torch.Generator().manual_seed(11)
    C = 16
    sc = [0.1, -0.05, 0.07, 0.02, -0.03, 0.04, 0.01, 0.9]
    w1 = torch.randn(C, C, 1, 1, generator=g) / C ** 0.5
    w2 = torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5
    w3 = torch.randn(C, C, 1, 1, generator=g) / C ** 0.5
    packed = [amd.ops.pack_conv_weight(w.cuda()) for w in (w1, w2, w3)]
    p = {"blk.branch_conv1.weight": w1, "blk.branch_conv2.weight": w2, "blk.branch_conv3.weight": w3}
    for n, v in zip(("bias1a", "bias1b", "bias2a", "bias2b", "bias3a", "bias3b", "bias4", "scale"), sc):
        p["blk." + n] = torch.tensor([v])
    for (B, H, W) in ((3, 16, 64), (1, 8, 32), (2, 40, 96)):
        x = torch.randn(B, H, W, C, generator=g).cuda()
        monkeypatch.setenv("VQAE_NO_WINO16", "1")
        a = amd.ops.fixup_same_block(x, *packed, sc)
        monkeypatch.delenv("VQAE_NO_WINO16")
        b = amd.ops.fixup_same_block(x, *packed, sc)
        ref = oracle.fixup_block(x.permute(0, 3, 1, 2).cpu(), p, "blk", "same").permute(0, 2, 3, 1)
        scale = float(ref.abs().max())
        e_ab, e_b = float((a - b).abs().max()) / scale, float((b.cpu() - ref).abs().max()) / scale
        record_parity("wino16_vs_direct", B=B, H=H, W=W, rel_err_vs_direct=e_ab, rel_err_vs_oracle=e_b)
        assert e_ab <= 2e-5 and e_b <= 2e-5, (B, H, W, e_ab, e_b)
