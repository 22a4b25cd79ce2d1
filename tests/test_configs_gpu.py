"""GPU: BASELINE.json configurations at their stated sizes.

configs[1]  batch 256 of 256x256x3, cfg B, fp32          -- full forward at batch 256
configs[2]  batch 256 of 512x512x3, cfg A, bf16          -- full forward at batch 256 (buffers of exactly 4 GiB)
configs[3]  256 / GPU of 256x256x3, cfg C, fp16          -- full forward at the per-GPU batch of the 8-GPU config
configs[4]  whole-slide extract_embeddings, cfg A encoder on 512x512 uint8 tiles -> HDF5

The oracle cannot run 256 patches of these models on the CPU in test time, so the full-size runs are checked through
size-independent properties: (i) batch invariance -- every patch is independent in eval mode, so
rows of the batch-256 result must equal, BIT FOR BIT, the same patches run in chunks of 2; (ii) the first rows of
the batch are the patches of the reference's fixture (recorded from the imported reference at batch 2 / 4), so those rows
carry the reference link at the full batch size.  A 32-bit offset wrap in any kernel at the 4 GiB buffer size breaks (i).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, record_parity

pytestmark = pytest.mark.gpu


def _params(oracle, name, g):
    p = oracle.make_params(oracle.SPECS[name], 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    return p


@pytest.mark.parametrize("name,tag,size,fixture", [("B", "f32", 256, "model_B"), ("A", "bf16", 512, "model_A_bf16"),
                                                   ("C", "f16", 256, "model_C_f16")])
def test_full_batch_256_batch_invariance_and_fixture_rows(amd, oracle, name, tag, size, fixture):
    g = load_golden(fixture)
    p = _params(oracle, name, g)
    nb = int(g["batch"])
    B = 256
    x = oracle.make_patches(B, size, 0)                       # rows [0, nb) are the fixture's patches (same generator stream)
    assert torch.equal(x[:nb], oracle.make_patches(nb, size, 0))
    nat = amd.NativeVQAE(amd.SPECS[name], p, compute_dtype=None if tag == "f32" else tag)
    xd = x.cuda()
    out, idx, loss = nat.forward(xd)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    # (ii) fixture rows inside the big batch
    ref = g["idx"].astype(np.int64)
    got = idx[:nb].cpu().numpy()
    agree = float((got == ref).mean())
    samp = out[:nb].cpu()[:, :, ::16, ::16]
    rel = float(((samp - torch.from_numpy(g["out_sample"])) ** 2).mean() / (torch.from_numpy(g["out_sample"]) ** 2).mean())
    # (i) batch invariance, bit for bit: chunks at the start, across the 2 GiB / 4 GiB byte offsets of the widest buffers, and the end
    bad = []
    for lo in (0, 62, 126, 128, 190, 254):
        o2, i2, _ = nat.forward(xd[lo:lo + 2].contiguous())
        if not (torch.equal(i2, idx[lo:lo + 2]) and torch.equal(o2, out[lo:lo + 2])):
            bad.append((lo, int((i2 != idx[lo:lo + 2]).sum()), float((o2 - out[lo:lo + 2]).abs().max())))
    record_parity("full_batch_256", config=name, dtype=tag, batch=B, fixture_rows=nb, fixture_idx_agreement=agree,
                  fixture_out_rel_mse=rel, invariance_failures=len(bad), loss=float(loss))
    assert not bad, bad
    if tag == "f32":
        clear = (g["second"] - g["best"]) > 2e-4 * g["second"]
        assert int(((got.reshape(-1) != ref.reshape(-1)) & clear).sum()) == 0 and agree >= 0.9995
        assert rel <= 1e-8
    else:
        assert agree >= (0.96 if tag == "bf16" else 0.99) and rel <= (5e-2 if tag == "bf16" else 1e-2)


def test_whole_slide_cfgA_512_tiles_to_hdf5(amd, oracle, tmp_path):
    """BASELINE configs[4] at its model / tile size (on one GPU, a small slide set): synthetic slides of 512x512 uint8
    tiles -> DataLoader -> device normalisation + cfg A encoder + projected VQ -> device-side stitching -> one HDF5 file
    (groups images / masks; convert.py:27-32), read back.  fp32 convolutions for the exact comparison with
    oracle-encoded tiles (a sampled subset: one CPU tile costs ~1 s), then the reference's default fp16 autocast for the
    whole set (>= 97 % of the fp32 codes, as the reference's own fp16 run is to its fp32 run)."""
    from vqae_amd import hdf5
    from vqae_amd.extract_embeddings import SyntheticSlideDataset, save_encodings_hdf5
    g = load_golden("model_A")
    spec = oracle.SPECS["A"]
    p = {k: v for k, v in _params(oracle, "A", g).items() if k.startswith("encoder.")}      # `del model.decoder` (:157)
    nat = amd.NativeVQAE(amd.SPECS["A"], p)
    ds = SyntheticSlideDataset([(3, 4), (2, 3)], patch_size=512, raw=True, names=["tumor_001", "normal_002"])
    out = save_encodings_hdf5(tmp_path / "slides.hdf5", nat, ds, batch_size=8, autocast_dtype=None, num_workers=4)
    r = hdf5.H5Reader(out)
    assert sorted(r.keys()) == ["images", "masks"]
    grids = {k: r["images"][k] for k in ("tumor_001", "normal_002")}
    assert grids["tumor_001"].shape == (96, 128) and grids["normal_002"].shape == (64, 96)
    assert grids["tumor_001"].dtype == np.uint8 and r["masks"]["tumor_001_mask"].shape == (96, 128)
    # sampled tiles against the oracle (fp32)
    checked, bad = 0, 0
    for index in (0, 7, 13):
        s, row, col = ds.locate(index)
        img, lab, _ = ds[index]
        (_,), (idx,), _ = oracle.encoder_forward(oracle.normalize_u8(img.numpy()[None]), p, spec)
        name = ["tumor_001", "normal_002"][s]
        tile = grids[name][row * 32:(row + 1) * 32, col * 32:(col + 1) * 32]
        bad += int((tile != idx[0].numpy()).sum())
        checked += 1024
        pooled = oracle.adaptive_max_pool_labels(lab.numpy(), 32)[0]
        assert np.array_equal(r["masks"][name + "_mask"][row * 32:(row + 1) * 32, col * 32:(col + 1) * 32].astype(np.uint8), pooled)
    # the default (reference) numerics: fp16 autocast
    out16 = save_encodings_hdf5(tmp_path / "slides_f16.hdf5", nat, ds, batch_size=8, num_workers=4)
    r16 = hdf5.H5Reader(out16)
    agree16 = float(np.mean([np.mean(r16["images"][k] == grids[k]) for k in grids]))
    record_parity("whole_slide_cfgA", tiles=len(ds), oracle_codes_checked=checked, oracle_mismatches=bad, f16_vs_f32_agreement=agree16)
    assert bad <= 2, bad                                         # near-tie rows only (conv summation order vs oneDNN)
    assert agree16 >= 0.97
