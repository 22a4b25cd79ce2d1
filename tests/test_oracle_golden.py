"""CPU: the oracle restatement replays the golden fixtures that tests/golden/make_golden.py
recorded from the imported reference (no GPU, no reference needed at test time)."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden, tap_sample


@pytest.mark.parametrize("D,K", [(8, 256), (128, 256), (256, 1024), (32, 16)])
def test_vq_oracle_matches_reference_fixture(oracle, D, K):
    g = load_golden(f"vq_D{D}_K{K}")
    N = int(g["N"])
    z, embed = oracle.make_vq_case(D, K, N, seed=int(g["seed"]))
    hw = 32 if N % 1024 == 0 else 16
    zin = z.reshape(N // (hw * hw), hw, hw, D).permute(0, 3, 1, 2).contiguous()
    q, idx, loss = oracle.vq_forward(zin, embed, 1.0)
    assert np.array_equal(idx.reshape(-1).numpy(), g["idx"].astype(np.int64))      # bit-exact indices
    assert float(loss) == float(g["loss"])
    assert np.array_equal(q.permute(0, 2, 3, 1).reshape(N, D)[::61].numpy(), g["q_flat_sample"])
    assert float(g["cdist_bitwise_match"]) == 1.0          # C restatement == torch.cdist, every entry


@pytest.mark.parametrize("tag", ["3d", "5d", "3d_wide"])
def test_vq_on_3d_and_5d_inputs_matches_reference_fixture(oracle, tag):
    """The reference passes p = inputs.dim() to torch.cdist (vq.py:97,121-129): [B, D, L] inputs are quantised under the 3-norm,
    [B, D, d, h, w] under the 5-norm.  Fixtures recorded from the reference's EMAVectorQuantizer.forward (make_golden.py::gen_vq_nd;
    the C restatement matched torch.cdist on 100 % of the entries there)."""
    g = load_golden(f"vq_nd_{tag}")
    D, K, N = int(g["D"]), int(g["K"]), int(g["N"])
    shape = tuple(int(v) for v in g["shape"])
    z, embed = oracle.make_vq_case(D, K, N, seed=int(g["seed"]))
    zin = z.reshape(*shape, D).permute(0, -1, *range(1, len(shape))).contiguous()
    q, idx, loss = oracle.vq_forward(zin, embed, 1.0)
    assert zin.dim() == len(shape) + 1 and tuple(idx.shape) == shape
    assert np.array_equal(idx.reshape(-1).numpy(), g["idx"].astype(np.int64))
    assert float(loss) == float(g["loss"])
    assert np.array_equal(q.reshape(q.shape[0], D, -1)[:, :, ::7].numpy(), g["q_sample"])
    assert float(g["cdist_bitwise_match"]) == 1.0


def test_vq_tie_rule_lowest_index(oracle):
    z, embed = oracle.make_vq_case(32, 16, 1024, seed=0)
    idx, best, second = oracle.vq_argmin_p4(z, embed)
    # rows 0..3 are exact copies of codes duplicated at K-1-i and i: the lower index must win
    assert idx[:4].tolist() == [0, 1, 2, 3]
    assert torch.all(best[:4] == 0) and torch.all(second[:4] == 0)


def test_vq_numpy_crosscheck(oracle):
    z, embed = oracle.make_vq_case(8, 256, 512, seed=1)
    idx, _, _ = oracle.vq_argmin_p4(z, embed)
    assert np.array_equal(oracle.vq_argmin_p4_numpy(z.numpy(), embed.numpy()), idx.numpy())


@pytest.mark.parametrize("name", ["tiny", "tinyP", "tinyM"])
def test_model_taps_match_reference_fixture(oracle, name):
    g = load_golden(f"model_{name}")
    spec = oracle.SPECS[name]
    fix, d = ast.literal_eval(str(g["spec"])), spec.to_dict()
    assert all(d[k] == v for k, v in fix.items())
    # fields added after a fixture was written must be at their defaults for that fixture's model
    assert all(d[k] == getattr(oracle.VQAESpec(), k) for k in d.keys() - fix.keys())
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    x = torch.from_numpy(g["x"])
    assert torch.equal(x, oracle.make_patches(int(g["batch"]), int(g["size"]), 0))
    taps = {}
    out, losses = oracle.vqae_forward(x, p, spec, taps)
    assert np.array_equal(taps["idx"].numpy(), g["idx"].astype(np.int64))
    assert float(losses[0]) == float(g["loss"])
    for k in g.files:
        if k.startswith("tap:") and k[4:] in taps:
            assert np.array_equal(taps[k[4:]].numpy(), g[k]), k
    assert np.array_equal(out.numpy(), g["tap:out"])


TDT = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}


@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", ["mid", "mid16", "midA", "midC", "midW"])
def test_block_taps_match_reference_fixture(oracle, name, tag):
    """Mid-size models (levels that reach the production kernels of cfg A/B/C): every block output of the oracle
    equals the reference's recorded one -- strided sample bit for bit, full tensor through its fp64 sum and sum of
    squares -- in fp32 and under CPU autocast bf16 / f16 (extract_embeddings.py:124-125)."""
    g = load_golden(f"taps_{name}_{tag}")
    spec = oracle.SPECS[name]
    assert ast.literal_eval(str(g["spec"])) == spec.to_dict()
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    x = oracle.make_patches(int(g["batch"]), int(g["size"]), 0)
    taps = {}
    out, losses = oracle.vqae_forward(x, p, spec, taps, dtype=TDT[tag])
    taps["out"] = out
    assert np.array_equal(taps["idx"].numpy(), g["idx"].astype(np.int64))
    n = 0
    for k in g.files:
        if not k.startswith("tap:"):
            continue
        t = taps[k[4:]].float()
        assert np.array_equal(tap_sample(t).numpy(), g[k]), k
        s1, s2 = g["sum:" + k[4:]]
        assert t.double().sum().item() == s1 and (t.double() ** 2).sum().item() == s2, k
        n += 1
    assert n == len(oracle.encoder_blocks(spec)) + len(oracle.decoder_blocks(spec)) + 4   # + stem, z, q, out


def test_model_B_matches_reference_fixture(oracle):
    """BASELINE config #1 (cfg B, batch 4, fp32, CPU): the oracle reproduces the reference's indices,
    loss and output samples exactly."""
    g = load_golden("model_B")
    spec = oracle.SPECS["B"]
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    x = oracle.make_patches(int(g["batch"]), int(g["size"]), 0)
    taps = {}
    out, losses = oracle.vqae_forward(x, p, spec, taps)
    assert np.array_equal(taps["idx"].numpy(), g["idx"].astype(np.int64))
    assert float(losses[0]) == float(g["loss"])
    assert np.array_equal(out[:, :, ::16, ::16].numpy(), g["out_sample"])
    assert abs(float(((out - x) ** 2).mean()) - float(g["recon_mse"])) < 1e-6 * float(g["recon_mse"])


def test_model_BM_matches_reference_fixture(oracle):
    """MBConv / EfficientNetV2 variant at cfg-B size (conf/model/{encoder,decoder}/efficientnetv2.yaml): the oracle's
    eval-mode MBConv restatement reproduces the imported reference exactly."""
    g = load_golden("model_BM")
    spec = oracle.SPECS["BM"]
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    x = oracle.make_patches(int(g["batch"]), int(g["size"]), 0)
    taps = {}
    out, losses = oracle.vqae_forward(x, p, spec, taps)
    assert np.array_equal(taps["idx"].numpy(), g["idx"].astype(np.int64))
    assert float(losses[0]) == float(g["loss"])
    assert np.array_equal(out[:, :, ::16, ::16].numpy(), g["out_sample"])


@pytest.mark.parametrize("name,tag,size", [("tiny", "f16", 32), ("tinyP", "bf16", 32), ("tinyM", "f16", 32), ("tinyM", "bf16", 32)])
def test_small_models_under_cpu_autocast_match_reference_fixture(oracle, name, tag, size):
    """The oracle under torch.autocast('cpu', dtype) -- the reference's extraction context (extract_embeddings.py:124-125) -- replays the
    indices, loss and output sample recorded from the reference itself under the same context; round 3 adds the MBConv variant
    (conv_block.py:240-321, layers/misc.py:23-30: BatchNorm, SiLU and the SE gate under autocast)."""
    import torch
    g = load_golden(f"model_{name}_{tag}")
    spec = oracle.SPECS[name]
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    x = oracle.make_patches(int(g["batch"]), size, 0)
    taps = {}
    out, losses = oracle.vqae_forward(x, p, spec, taps, dtype={"f16": torch.float16, "bf16": torch.bfloat16}[tag])
    assert np.array_equal(taps["idx"].numpy().astype(np.int64), g["idx"].astype(np.int64))
    assert np.array_equal(out.float()[:, :, ::16, ::16].numpy(), g["out_sample"])


def test_se_hidden_is_make_divisible(oracle):
    # utils/train_helpers.py:21-24 with divide=True
    assert [oracle.se_hidden(c, 4) for c in (32, 64, 128, 512, 1024, 6)] == [8, 16, 32, 128, 256, 2]


def test_bicubic_explicit_restatement(oracle):
    x = torch.randn(2, 8, 9, 7)
    a, b = oracle.bicubic_up2(x), oracle.bicubic_up2_explicit(x)
    assert (a - b).abs().max() <= 4 * torch.finfo(torch.float32).eps * x.abs().max()


def test_driver_fixture(oracle):
    g = load_golden("driver")
    tiles, meta, sizes = g["tiles"], g["meta"], g["sizes"]
    for s, name in enumerate(["images/slide_a", "images/slide_b"]):
        sel = tiles[meta[:, 0] == s]
        grid = oracle.cast_to_lowest_dtype(oracle.stitch_slide(sel, *sizes[s]))
        ref = g["grid:" + name]
        assert grid.dtype == ref.dtype and np.array_equal(grid, ref)
    assert g["grid:images/slide_b"].dtype == np.bool_
    assert g["grid:images/slide_a"].dtype == np.uint8


def test_ema_fixture(oracle):
    g = load_golden("ema")
    D, K = int(g["D"]), int(g["K"])
    z0, embed = oracle.make_vq_case(D, K, 1024, seed=3, adversarial=False)
    z1, _ = oracle.make_vq_case(D, K, 1024, seed=4, adversarial=False)
    e, ea, cs = oracle.init_ema(z0 * 1.7 + 0.3, embed, embed.clone(), torch.zeros(K))
    for step, z in enumerate((z0 * 1.7 + 0.3, z1 * 1.7 + 0.3)):
        idx, _, _ = oracle.vq_argmin_p4(z, e)
        assert np.array_equal(idx.numpy(), g[f"idx{step}"].astype(np.int64))
        e, ea, cs = oracle.update_ema(z, idx, ea, cs, 0.99, 1e-5)
        np.testing.assert_allclose(e.numpy(), g[f"embed{step}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ea.numpy(), g[f"embed_avg{step}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cs.numpy(), g[f"cluster_size{step}"], rtol=1e-6, atol=1e-7)


def test_label_maxpool_matches_torch(oracle):
    lab = (torch.rand(3, 64, 64) > 0.97).to(torch.uint8)
    ref = torch.nn.functional.adaptive_max_pool2d(lab[:, None].half().float(), 32)[:, 0].to(torch.uint8)
    assert np.array_equal(oracle.adaptive_max_pool_labels(lab.numpy(), 32), ref.numpy())
