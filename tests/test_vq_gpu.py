"""GPU parity: the HIP vector quantiser (through the C ABI) against the CPU oracle.
Bar: code indices bit-exact on identical inputs; q bit-exact; loss within 1e-6 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def run_hip(amd, z, embed, **kw):
    q, idx, loss, margin = amd.ops.vq_forward(z.cuda(), embed.cuda(), 1.0, want_margin=True, **kw)
    torch.cuda.synchronize()
    return q.cpu(), idx.cpu(), float(loss), margin.cpu()


@pytest.mark.parametrize("D,K", [(8, 256), (128, 256), (256, 1024), (32, 16)])
def test_vq_matches_golden_and_oracle(amd, oracle, D, K):
    g = load_golden(f"vq_D{D}_K{K}")
    N = int(g["N"])
    z, embed = oracle.make_vq_case(D, K, N, seed=int(g["seed"]))
    q, idx, loss, margin = run_hip(amd, z, embed)
    ref = g["idx"].astype(np.int64)
    assert np.array_equal(idx.numpy(), ref), f"{(idx.numpy() != ref).sum()} index mismatches"
    # q = z + (e[idx] - z), bitwise (vq.py:146)
    e = embed[idx]
    assert torch.equal(q, z + (e - z))
    assert np.array_equal(q[::61].numpy(), g["q_flat_sample"])
    assert abs(loss - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    # adversarial rows: exact duplicates -> margin 0 -> resolved to the lowest index
    assert idx[:4].tolist() == [0, 1, 2, 3]
    assert torch.all(margin[:4] == 0)


@pytest.mark.parametrize("N", [1, 7, 127, 128, 129, 1000])
@pytest.mark.parametrize("D,K", [(8, 256), (128, 256), (64, 300), (12, 5)])
def test_vq_ragged_sizes(amd, oracle, N, D, K):
    z, embed = oracle.make_vq_case(D, K, max(N, 128), seed=5, adversarial=K >= 32)
    z = z[:N].contiguous()
    q, idx, loss, _ = run_hip(amd, z, embed)
    oidx, _, _ = oracle.vq_argmin_p4(z, embed)
    assert torch.equal(idx, oidx)
    ref_loss = float(((z - embed[oidx]) ** 2).double().mean())
    assert abs(loss - ref_loss) <= 2e-6 * max(ref_loss, 1e-30)


@pytest.mark.parametrize("N,D,K", [(1000, 6, 40), (257, 13, 300), (64, 3, 5), (130, 130, 64)])
def test_vq_any_embedding_dim(amd, oracle, N, D, K):
    """embedding_dim not a multiple of 4 (the reference accepts any, vq.py:121-129): zero-padded copies inside the call;
    indices / q identical to the oracle's, loss a mean over the REAL N * D elements; embed_code on such a codebook."""
    z, embed = oracle.make_vq_case(D, K, max(N, 128), seed=9, adversarial=K >= 32)
    z = z[:N].contiguous()
    q, idx, loss, margin = run_hip(amd, z, embed)
    oidx, _, _ = oracle.vq_argmin_p4(z, embed)
    assert torch.equal(idx, oidx)
    assert torch.equal(q, z + (embed[oidx] - z))
    ref_loss = float(((z - embed[oidx]) ** 2).double().mean())
    assert abs(loss - ref_loss) <= 2e-6 * max(ref_loss, 1e-30)
    got = amd.ops.embed_code(oidx.cuda(), embed.cuda()).cpu()
    assert torch.equal(got, embed[oidx])


@pytest.mark.parametrize("tag", ["3d", "5d", "3d_wide"])
def test_vq_on_3d_and_5d_inputs(amd, oracle, tag):
    """p = inputs.dim() (vq.py:97,121-129): the kernel under the 3- / 5-norm and the module mirror on [B, D, L] / [B, D, d, h, w]
    inputs against the fixture recorded from the reference (indices bit-exact incl. the duplicate-code / near-tie rows, q bit-exact
    on the sample, loss <= 1e-6 relative)."""
    from vqae_amd.layers.vq import EMAVectorQuantizer
    g = load_golden(f"vq_nd_{tag}")
    D, K, N = int(g["D"]), int(g["K"]), int(g["N"])
    shape = tuple(int(v) for v in g["shape"])
    nd = len(shape) + 1
    z, embed = oracle.make_vq_case(D, K, N, seed=int(g["seed"]))
    ref = g["idx"].astype(np.int64)
    q, idx, loss, margin = amd.ops.vq_forward(z.cuda(), embed.cuda(), 1.0, want_margin=True, p=nd)
    assert np.array_equal(idx.cpu().numpy(), ref), f"{(idx.cpu().numpy() != ref).sum()} index mismatches (p = {nd})"
    assert torch.equal(q.cpu(), z + (embed[idx.cpu()] - z))
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    assert idx[:4].tolist() == [0, 1, 2, 3] and bool(torch.all(margin[:4].cpu() == 0))      # adversarial duplicates -> lowest index
    m = EMAVectorQuantizer(K, D, 1.0, 0.99, 1e-5).eval()
    with torch.no_grad():
        m.embed.copy_(embed)
    m = m.cuda()
    zin = z.reshape(*shape, D).permute(0, -1, *range(1, len(shape))).contiguous()
    qm, im, lm = m(zin.cuda())
    assert qm.shape == zin.shape and im.shape == shape and im.dtype == torch.int64
    assert np.array_equal(im.reshape(-1).cpu().numpy(), ref)
    assert np.array_equal(qm.reshape(qm.shape[0], D, -1)[:, :, ::7].cpu().numpy(), g["q_sample"])
    assert abs(float(lm) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))


def test_vq_input_rank_errors(amd):
    """Reference error behaviour: AssertionError for ndim < 3 (vq.py:98), NotImplementedError for a wrong channel count
    (vq.py:100-104); ranks above 5 (p > 5) are not implemented here."""
    from vqae_amd.layers.vq import EMAVectorQuantizer
    m = EMAVectorQuantizer(8, 4, 1.0, 0.99, 1e-5).eval().cuda()
    with pytest.raises(AssertionError):
        m(torch.zeros(3, 4).cuda())
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 5, 7).cuda())
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 4, 2, 2, 2, 2).cuda())


def test_vq_empty(amd):
    q, idx, loss, _ = amd.ops.vq_forward(torch.zeros(0, 8).cuda(), torch.randn(4, 8).cuda(), want_margin=True)
    torch.cuda.synchronize()
    assert idx.numel() == 0 and float(loss) == 0.0


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32, torch.uint8])
def test_vq_index_dtypes(amd, oracle, dtype):
    z, embed = oracle.make_vq_case(16, 200, 512, seed=2)
    _, idx, _, _ = run_hip(amd, z, embed, idx_dtype=dtype)
    oidx, _, _ = oracle.vq_argmin_p4(z, embed)
    assert idx.dtype == dtype and torch.equal(idx.to(torch.int64), oidx)


def test_vq_full_size_properties(amd, oracle):
    """BASELINE config #2 size (262,144 rows, K=256, D=128): the oracle needs ~10 s for all rows on
    16 cores, so check size-independent properties on every row and the oracle on a 1/32 sample."""
    N, D, K = 256 * 1024, 128, 256
    g = torch.Generator().manual_seed(11)
    embed = torch.randn(K, D, generator=g)
    z = torch.randn(N, D, generator=g)
    z[:K] = embed                                   # idempotence: codes map to themselves
    zc, ec = z.cuda(), embed.cuda()
    q, idx, loss, margin = amd.ops.vq_forward(zc, ec, 1.0, want_margin=True)
    torch.cuda.synchronize()
    assert torch.equal(idx[:K].cpu(), torch.arange(K))
    assert torch.equal(q, zc + (ec[idx] - zc))     # lookup + straight-through, every row
    # chosen code is no farther (p=4 sums, fp64) than 64 random other codes, every row
    d_best = ((zc - ec[idx]).double() ** 4).sum(1)
    for t in range(4):
        other = torch.randint(0, K, (N,), device="cuda", generator=None)
        d_o = ((zc - ec[other]).double() ** 4).sum(1)
        assert bool((d_best <= d_o * (1 + 1e-6)).all())
    sel = torch.arange(0, N, 32)
    oidx, _, _ = oracle.vq_argmin_p4(z[sel], embed)
    assert torch.equal(idx.cpu()[sel], oidx)
    ref_loss = float(((zc - ec[idx]).double() ** 2).mean())
    assert abs(float(loss) - ref_loss) <= 2e-6 * ref_loss


def test_embed_code(amd):
    embed = torch.randn(37, 24)
    idx = torch.randint(0, 37, (5, 6, 7))
    out = amd.ops.embed_code(idx.cuda(), embed.cuda()).cpu()
    assert torch.equal(out, torch.nn.functional.embedding(idx, embed))


def test_ema_bookkeeping_matches_golden(amd, oracle):
    """Training-mode VQ (vq.py:47-94) against the reference-recorded fixture."""
    from vqae_amd.layers.vq import EMAVectorQuantizer
    g = load_golden("ema")
    D, K = int(g["D"]), int(g["K"])
    z0, embed = oracle.make_vq_case(D, K, 1024, seed=3, adversarial=False)
    z1, _ = oracle.make_vq_case(D, K, 1024, seed=4, adversarial=False)
    vq = EMAVectorQuantizer(K, D, 1.0, 0.99, 1e-5).cuda().train()
    vq.embed.copy_(embed.cuda()); vq.embed_avg.copy_(embed.cuda())
    for step, z in enumerate((z0 * 1.7 + 0.3, z1 * 1.7 + 0.3)):
        zin = z.reshape(1, 32, 32, D).permute(0, 3, 1, 2).contiguous().cuda()
        q, idx, loss = vq(zin)
        torch.cuda.synchronize()
        assert np.array_equal(idx.reshape(-1).cpu().numpy(), g[f"idx{step}"].astype(np.int64))
        np.testing.assert_allclose(vq.embed.cpu().numpy(), g[f"embed{step}"], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(vq.embed_avg.cpu().numpy(), g[f"embed_avg{step}"], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(vq.cluster_size.cpu().numpy(), g[f"cluster_size{step}"], rtol=2e-6, atol=1e-6)
        assert abs(float(loss) - float(g[f"loss{step}"])) <= 2e-6 * float(g[f"loss{step}"])
    assert int(vq.first_pass) == 0


# ---- fused projected quantiser (projection_dim = 8, the reference default): vqae_vq_projected_f32 --------------------
def _proj_case(oracle, C, K, N, seed):
    g = torch.Generator().manual_seed(seed)
    p = {"vq.proj_in.weight": torch.randn(8, C, 1, 1, generator=g) / C ** 0.5, "vq.proj_in.bias": torch.randn(8, generator=g) * 0.1,
         "vq.proj_out.weight": torch.randn(C, 8, 1, 1, generator=g) / 8 ** 0.5, "vq.proj_out.bias": torch.randn(C, generator=g) * 0.1,
         "vq.embed": torch.randn(K, 8, generator=g)}
    hw = int(N ** 0.5)
    x = torch.randn(1, C, hw, N // hw, generator=g)
    return p, x


def test_vq_projected_index_exact_on_identical_z(amd, oracle):
    """proj_in = a channel selection (z_j = x[c_j], exact in any summation order), so the fused kernel quantises
    exactly the rows of the reference-recorded D = 8 fixture: indices must be bit-exact (incl. the duplicate-code /
    near-tie rows that go through tier 2), q and the loss as the plain kernel's, out = proj_out(q)."""
    g = load_golden("vq_D8_K256")
    N, C = int(g["N"]), 128
    z, embed = oracle.make_vq_case(8, 256, N, seed=int(g["seed"]))
    sel = [3, 17, 29, 45, 64, 90, 101, 127]
    w_in = torch.zeros(8, C)
    for j, c in enumerate(sel):
        w_in[j, c] = 1.0
    gen = torch.Generator().manual_seed(1)
    x = torch.zeros(N, C)
    x[:, sel] = z
    w_out, b_out = torch.randn(C, 8, generator=gen), torch.randn(C, generator=gen)
    out, idx, loss, zz, margin = amd.ops.vq_projected(x.cuda(), w_in.cuda(), torch.zeros(8).cuda(), embed.cuda(), w_out.cuda(),
                                                      b_out.cuda(), want_z=True, want_margin=True)
    torch.cuda.synchronize()
    assert torch.equal(zz.cpu(), z)
    ref = g["idx"].astype(np.int64)
    assert np.array_equal(idx.cpu().numpy(), ref), f"{(idx.cpu().numpy() != ref).sum()} index mismatches"
    assert idx[:4].tolist() == [0, 1, 2, 3] and bool(torch.all(margin[:4].cpu() == 0))
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    q = z + (embed[idx.cpu()] - z)
    want = torch.nn.functional.linear(q, w_out, b_out)
    assert float((out.cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    # without the margin output the search runs behind the f16 matrix-pipe filter (vq_proj16_kernel): same indices, bit for bit
    out2, idx2, loss2, zz2, _ = amd.ops.vq_projected(x.cuda(), w_in.cuda(), torch.zeros(8).cuda(), embed.cuda(), w_out.cuda(),
                                                     b_out.cuda(), want_z=True)
    torch.cuda.synchronize()
    assert torch.equal(zz2.cpu(), z)
    assert np.array_equal(idx2.cpu().numpy(), ref), f"filter path: {(idx2.cpu().numpy() != ref).sum()} index mismatches"
    assert torch.equal(out2, out) and float(loss2) == float(loss)


@pytest.mark.parametrize("N,K,scale", [(4096, 256, 1.0), (1000, 64, 1.0), (129, 256, 40.0), (2048, 256, 1e-3), (1, 32, 1.0), (70, 240, 7.0),
                                       (4096, 250, 1.0), (512, 512, 1.0)])
def test_vq_projected_filter_path_index_exact(amd, oracle, N, K, scale):
    """C = 128 without the margin output takes vq_proj16_kernel's filter (one f16 MFMA per 16 rows x 16 codes -> survivors ->
    exact tier-1 evaluation -> tier 2 for near ties; K > 256 or K % 16 != 0: every code exactly).  proj_in is a channel
    selection, so z is known exactly and the indices must equal the oracle's on adversarial rows (duplicated codes, exact
    hits, 0-4 ulp near ties), ragged N, 40x / 0.001x magnitudes and rows outside the f16 range of z^3."""
    C = 128
    z, embed = oracle.make_vq_case(8, K, max(N, 128), seed=5, adversarial=K >= 32)
    z = (z[:N] * scale).contiguous()
    embed = (embed * scale).contiguous()
    if N >= 64:
        z[5] *= 1000.0                                   # leaves the filter's range: the wave step scans every code
        z[17] = embed[K // 2] + 1e-4 * scale
        z[33] = 0.0
    sel = [3, 17, 29, 45, 64, 90, 101, 127]
    w_in = torch.zeros(8, C)
    for j, c in enumerate(sel):
        w_in[j, c] = 1.0
    x = torch.zeros(N, C)
    x[:, sel] = z
    gen = torch.Generator().manual_seed(2)
    w_out, b_out = torch.randn(C, 8, generator=gen), torch.randn(C, generator=gen)
    out, idx, loss, zz, _ = amd.ops.vq_projected(x.cuda(), w_in.cuda(), torch.zeros(8).cuda(), embed.cuda(), w_out.cuda(), b_out.cuda(),
                                                 want_z=True)
    torch.cuda.synchronize()
    assert torch.equal(zz.cpu(), z)
    oidx, _, _ = oracle.vq_argmin_p4(z, embed)
    bad = (idx.cpu() != oidx).nonzero().flatten().tolist()
    assert not bad, (bad[:10], idx.cpu()[bad[:10]].tolist(), oidx[bad[:10]].tolist())
    q = z + (embed[oidx] - z)
    want = torch.nn.functional.linear(q, w_out, b_out)
    assert float((out.cpu() - want).abs().max()) <= 2e-6 * max(1e-30, float(want.abs().max()))


@pytest.mark.parametrize("N,C,K", [(4096, 128, 256), (1000, 128, 64), (1, 64, 32), (130, 256, 300)])
def test_vq_projected_matches_oracle(amd, oracle, N, C, K):
    """Random projections: against ProjectedEMAVectorQuantizer2d's oracle restatement (vq.py:190-192).  z differs from
    oneDNN's by fp32 summation order, so indices must agree on every row whose margin exceeds that noise."""
    hw = {4096: (64, 64), 1000: (25, 40), 1: (1, 1), 130: (10, 13)}[N]
    g = torch.Generator().manual_seed(N + C)
    p, _ = _proj_case(oracle, C, K, N, N + C)
    x = torch.randn(1, C, *hw, generator=g)
    want, oidx, oloss = oracle.projected_vq_forward(x, p, "vq.")
    xf = x.permute(0, 2, 3, 1).reshape(N, C)
    out, idx, loss, z, margin = amd.ops.vq_projected(xf.cuda(), p["vq.proj_in.weight"].cuda(), p["vq.proj_in.bias"].cuda(),
                                                     p["vq.embed"].cuda(), p["vq.proj_out.weight"].cuda(), p["vq.proj_out.bias"].cuda(),
                                                     want_z=True, want_margin=True)
    torch.cuda.synchronize()
    zo = torch.nn.functional.conv2d(x, p["vq.proj_in.weight"], p["vq.proj_in.bias"]).permute(0, 2, 3, 1).reshape(N, 8)
    assert float((z.cpu() - zo).abs().max()) <= 2e-6 * max(1.0, float(zo.abs().max()))
    same = idx.cpu() == oidx.reshape(-1)
    clear = margin.cpu() > 1e-4
    assert bool(same[clear].all()) and float(same.float().mean()) >= 0.999
    wf = want.permute(0, 2, 3, 1).reshape(N, C)
    assert float((out.cpu() - wf)[same].abs().max()) <= 1e-5 * float(wf.abs().max())
    assert abs(float(loss) - float(oloss)) <= 1e-4 * float(oloss)


@pytest.mark.parametrize("tag", ["bf16", "f16"])
def test_vq_projected_autocast(amd, oracle, tag):
    """Under torch.autocast the two projections are 16-bit convolutions (fp32 distance / q / loss): vs the oracle
    under CPU autocast."""
    N, C, K = 4096, 128, 256
    p, x = _proj_case(oracle, C, K, N, 7)
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[tag]
    with torch.autocast("cpu", dtype=dt):
        want, oidx, oloss = oracle.projected_vq_forward(x, p, "vq.")
    xf = x.permute(0, 2, 3, 1).reshape(N, C)
    out, idx, loss, _, _ = amd.ops.vq_projected(xf.cuda(), p["vq.proj_in.weight"].cuda(), p["vq.proj_in.bias"].cuda(),
                                                p["vq.embed"].cuda(), p["vq.proj_out.weight"].cuda(), p["vq.proj_out.bias"].cuda(), dtype=tag)
    torch.cuda.synchronize()
    same = idx.cpu() == oidx.reshape(-1)
    assert float(same.float().mean()) >= 0.995                    # a 16-bit rounding flip of z moves near-tie rows
    wf = want.float().permute(0, 2, 3, 1).reshape(N, C)
    ulp = 2.0 ** (-8 if tag == "bf16" else -11)
    assert float((out.cpu() - wf)[same].abs().max()) <= 2 * ulp * float(wf.abs().max())


def test_vq_projected_fused_equals_unfused_in_the_handle(amd, oracle, monkeypatch):
    """cfg-A-style model (midA): handle with the fused quantiser vs VQAE_NO_VQ_FUSION=1 (proj_in conv -> VQ -> proj_out conv)."""
    spec = oracle.SPECS["midA"]
    p = oracle.make_params(spec, 0)
    x = oracle.make_patches(4, 128, 3)
    p = oracle.calibrate_codebook(x[:2], p, spec)
    fused = amd.NativeVQAE(amd.SPECS["midA"], p)
    monkeypatch.setenv("VQAE_NO_VQ_FUSION", "1")
    plain = amd.NativeVQAE(amd.SPECS["midA"], p)
    of, idf, lf = fused.forward(x.cuda())
    op, idp, lp = plain.forward(x.cuda())
    torch.cuda.synchronize()
    assert float((idf == idp).float().mean()) >= 0.999
    assert abs(float(lf) - float(lp)) <= 1e-5 * float(lp)
    if bool((idf == idp).all()):
        assert float((of - op).abs().max()) <= 1e-4 * float(op.abs().max())


@pytest.mark.parametrize("N,K,scale", [(4096, 1024, 1.0), (1000, 300, 1.0), (129, 1024, 40.0), (2048, 64, 1e-3), (1, 32, 1.0), (70, 1000, 7.0)])
def test_vq_filter_path_matches_oracle(amd, oracle, N, K, scale):
    """D = 256 without the margin output takes the matrix-pipe filter (csrc/vq_filter.hip: f16 GEMM scores -> survivors ->
    exact fp32 evaluation -> tier 2 for near ties).  Index-exact against the oracle on: duplicate / near-tie rows (adversarial
    cases), K not a multiple of the 128-code padding, ragged N, codebooks and rows 40x and 1/1000 the usual magnitude
    (the operands are rescaled by 8 / max|e|), and rows far outside the f16 range of z^3 (they take every code)."""
    D = 256
    z, embed = oracle.make_vq_case(D, K, max(N, 128), seed=11, adversarial=K >= 32)
    z = (z[:N] * scale).contiguous()
    embed = (embed * scale).contiguous()
    if N >= 64:
        z[5] *= 1000.0                                   # |z| * 8 / max|e| >> 30: the row leaves the filter's range
        z[17] = embed[K // 2] + 1e-4 * scale             # a row next to a code
        z[33] = 0.0
    q, idx, loss, margin = amd.ops.vq_forward(z.cuda(), embed.cuda(), 1.0, want_margin=False)
    torch.cuda.synchronize()
    assert margin is None
    oidx, _, _ = oracle.vq_argmin_p4(z, embed)
    bad = (idx.cpu() != oidx).nonzero().flatten().tolist()
    assert not bad, (bad[:10], idx.cpu()[bad[:10]].tolist(), oidx[bad[:10]].tolist())
    assert torch.equal(q.cpu(), z + (embed[oidx] - z))


def test_vq_filter_path_golden_and_switch(amd, oracle, monkeypatch):
    """The reference's golden indices (D = 256, K = 1024) through the filter path, and the same call with the filter switched
    off (VQAE_NO_VQ_FILTER is read once per process, so the comparison is with the margin-producing call, which always takes
    vq_tier1_kernel): identical indices, loss and q."""
    g = load_golden("vq_D256_K1024")
    z, embed = oracle.make_vq_case(256, 1024, int(g["N"]), seed=int(g["seed"]))
    qf, idxf, lossf, _ = amd.ops.vq_forward(z.cuda(), embed.cuda(), 1.0, want_margin=False)
    qt, idxt, losst, _ = amd.ops.vq_forward(z.cuda(), embed.cuda(), 1.0, want_margin=True)
    torch.cuda.synchronize()
    assert np.array_equal(idxf.cpu().numpy(), g["idx"].astype(np.int64))
    assert torch.equal(idxf, idxt) and torch.equal(qf, qt) and float(lossf) == float(losst)
    assert idxf[:4].tolist() == [0, 1, 2, 3]            # exact duplicates resolve to the lowest index
