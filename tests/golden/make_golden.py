#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference modules (imported from
/root/reference through tests/golden/_ref_shims.py) on seeded synthetic inputs, and check the
oracle restatement (oracle/vqae_oracle.py, oracle/vq_p4.c) against them while doing so.

Runs only in the build container (the reference does not exist on the GPU box).  The fixtures are
data: inputs are regenerated from seeds by oracle.vqae_oracle.make_* on both sides; the files hold
the reference's outputs (indices, losses, samples of activations), never reference source.

    python tests/golden/make_golden.py [--only vq,vqnd,tiny,B,A,C,driver,ema]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_shims as S  # noqa: E402
from oracle import vqae_oracle as O  # noqa: E402

torch.set_grad_enabled(False)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------- G1: VQ kernel cases
def gen_vq():
    S.install()
    from vq_ae.layers.vq import EMAVectorQuantizer  # the reference
    for D, K, N in ((8, 256, 4096), (128, 256, 4096), (256, 1024, 2048), (32, 16, 1024)):
        z, embed = O.make_vq_case(D, K, N, seed=0)
        vq = EMAVectorQuantizer(num_embeddings=K, embedding_dim=D, commitment_cost=1.0, decay=0.99,
                                laplace_alpha=1e-5).eval()
        vq.embed.copy_(embed)
        # reference contract: inputs [B, D, h, w]; N = B*h*w with h*w = 32*32 (or 16*16)
        hw = 32 if N % 1024 == 0 else 16
        B = N // (hw * hw)
        zin = z.reshape(B, hw, hw, D).permute(0, 3, 1, 2).contiguous()
        t = time.time()
        q, idx, loss = vq(zin)
        t_ref = time.time() - t
        # pin the C restatement bit-for-bit against torch.cdist itself
        dist = torch.cdist(z, embed, 4.0, compute_mode="donot_use_mm_for_euclid_dist")
        import ctypes
        lib = O._c_lib()
        mine = torch.empty_like(dist)
        lib.vq_p4_cdist_ref(z.data_ptr(), embed.data_ptr(), N, K, D, ctypes.c_float(4.0), mine.data_ptr(),
                            os.cpu_count())
        match = (mine == dist).float().mean().item()
        oidx, best, second = O.vq_argmin_p4(z, embed, 4.0)
        assert torch.equal(oidx, idx.reshape(-1)), "C oracle argmin != reference"
        oq, oi, ol = O.vq_forward(zin, embed, 1.0)
        assert torch.equal(oq, q) and torch.equal(oi, idx) and float(ol) == float(loss)
        if N * K * D <= 2 ** 24:
            assert np.array_equal(O.vq_argmin_p4_numpy(z.numpy(), embed.numpy()), idx.reshape(-1).numpy())
        print(f"  vq D={D} K={K} N={N}: ref {t_ref:.2f}s, C-oracle cdist bitwise match {match * 100:.4f}%, "
              f"argmin equal, min margin {(second - best).min().item():.3e}")
        save(f"vq_D{D}_K{K}", D=D, K=K, N=N, seed=0, idx=idx.reshape(-1).numpy().astype(np.uint16),
             loss=np.float32(loss.item()), q_flat_sample=q.permute(0, 2, 3, 1).reshape(N, D)[::61].numpy(),
             best=best.numpy(), second=second.numpy(), cdist_bitwise_match=np.float64(match))


def gen_vq_nd():
    """EMAVectorQuantizer.forward on 3-D and 5-D inputs: the reference passes p = inputs.dim() to torch.cdist (vq.py:97,121-129),
    so [B, D, L] is quantised under the 3-norm and [B, D, d, h, w] under the 5-norm.  Adversarial rows as in the 4-D cases."""
    S.install()
    from vq_ae.layers.vq import EMAVectorQuantizer  # the reference
    import ctypes
    for tag, D, K, shape in (("3d", 16, 64, (4, 256)), ("5d", 12, 40, (2, 4, 8, 16)), ("3d_wide", 128, 256, (2, 512))):
        N = int(np.prod(shape))
        nd = len(shape) + 1
        z, embed = O.make_vq_case(D, K, N, seed=3)
        vq = EMAVectorQuantizer(num_embeddings=K, embedding_dim=D, commitment_cost=1.0, decay=0.99, laplace_alpha=1e-5).eval()
        vq.embed.copy_(embed)
        zin = z.reshape(*shape, D).permute(0, -1, *range(1, len(shape))).contiguous()      # [B, D, *spatial]
        assert zin.dim() == nd
        q, idx, loss = vq(zin)
        dist = torch.cdist(z, embed, float(nd), compute_mode="donot_use_mm_for_euclid_dist")
        mine = torch.empty_like(dist)
        O._c_lib().vq_p4_cdist_ref(z.data_ptr(), embed.data_ptr(), N, K, D, ctypes.c_float(float(nd)), mine.data_ptr(), os.cpu_count())
        match = (mine == dist).float().mean().item()
        oq, oi, ol = O.vq_forward(zin, embed, 1.0)
        assert torch.equal(oi, idx), "oracle argmin != reference"
        assert torch.equal(oq, q) and float(ol) == float(loss)
        oidx, best, second = O.vq_argmin_p4(z, embed, float(nd))
        print(f"  vq {tag}: rank-{nd} input {tuple(zin.shape)} (p = {nd}), C-oracle cdist bitwise match {match * 100:.4f}%, argmin equal, "
              f"min margin {(second - best).min().item():.3e}")
        save(f"vq_nd_{tag}", D=D, K=K, N=N, seed=3, shape=np.asarray(shape), idx=idx.reshape(-1).numpy().astype(np.uint16),
             loss=np.float32(loss.item()), q_sample=q.reshape(q.shape[0], D, -1)[:, :, ::7].numpy(), cdist_bitwise_match=np.float64(match))


# ---------------------------------------------------------------- G2..G4: model cases
def ref_forward_with_taps(model, x):
    """Run the reference VQAE block by block to record intermediates (same modules, same order as
    Encoder.forward model.py:198-208 / Decoder.forward :278-291 for the single-level case)."""
    taps = {}
    enc, dec = model.encoder, model.decoder
    h = enc.in_stem(x)
    taps["stem"] = h
    for lvl, env in enumerate(enc.down_layers[0].layers):
        for b, blk in enumerate(env.layers):
            h = blk(h)
            taps[f"encoder.down_layers.0.layers.{lvl}.layers.{b}"] = h
    for i, blk in enumerate(enc.pre_enc_layers[0]):
        h = blk(h)
        taps[f"encoder.pre_enc_layers.0.{i}"] = h
    taps["z"] = h
    q, idx, loss = enc.vq_layers[0](h)
    taps["q"], taps["idx"], taps["loss"] = q, idx, loss
    h = q
    for i, blk in enumerate(dec.post_enc_layers[0]):
        h = blk(h)
        taps[f"decoder.post_enc_layers.0.{i}"] = h
    for lvl, env in enumerate(dec.up_layers[0].layers):
        for b, blk in enumerate(env.layers):
            h = blk(h)
            taps[f"decoder.up_layers.0.layers.{lvl}.layers.{b}"] = h
    taps["out"] = dec.out_stem(h)
    return taps


def gen_model(name, batch, size, full_taps):
    spec = O.SPECS[name]
    p = O.make_params(spec, 0)
    x = O.make_patches(batch, size, 0)
    p = O.calibrate_codebook(O.make_patches(2, size, 99), p, spec)
    model = S.build_reference_model(spec, p)
    t = time.time()
    out, losses = model(x)                                   # VQAE.forward, model.py:41-48
    (q,), (idx,), (loss,) = model.encoder(x)                 # Encoder.forward contract
    t_ref = time.time() - t
    rt = ref_forward_with_taps(model, x)
    assert torch.equal(rt["out"], out) and torch.equal(rt["idx"], idx)
    ot = {}
    oout, olosses = O.vqae_forward(x, p, spec, ot)
    same_idx = (ot["idx"] == idx).float().mean().item()
    dout = (oout - out).abs().max().item()
    print(f"  {name}: reference fwd+enc {t_ref:.1f}s; oracle idx agreement {same_idx * 100:.4f}%, "
          f"out max|diff| {dout:.3e}, loss ref {loss.item():.6f} oracle {olosses[0].item():.6f}, "
          f"codes used {idx.unique().numel()}/{spec.num_embeddings}")
    assert same_idx == 1.0 and dout == 0.0, "oracle restatement deviates from the reference"
    # margins of the reference's own z (for tie-aware end-to-end comparisons)
    zr = rt["z"]
    vq = "encoder.vq_layers.0."
    zc = torch.nn.functional.conv2d(zr, p[vq + "proj_in.weight"], p[vq + "proj_in.bias"]) \
        if spec.projection_dim > 0 else zr
    flat = zc.permute(0, 2, 3, 1).reshape(-1, zc.shape[1])
    oidx, best, second = O.vq_argmin_p4(flat, p[vq + "embed"], 4.0)
    assert torch.equal(oidx.reshape(idx.shape), idx)
    arrays = dict(
        spec=np.array(repr(spec.to_dict())), batch=batch, size=size,
        idx=idx.numpy().astype(np.uint16), loss=np.float32(loss.item()),
        recon_mse=np.float64(((out - x) ** 2).mean().item()),
        out_mean=out.mean(dim=(1, 2, 3)).numpy(), out_std=out.std(dim=(1, 2, 3)).numpy(),
        out_sample=out[:, :, ::16, ::16].numpy(), z_sample=zr[:, ::8, ::4, ::4].numpy(),
        q_sample=q[:, ::8, ::4, ::4].numpy(),
        best=best.numpy(), second=second.numpy(),
        embed=p[vq + "embed"].numpy(),   # calibrated codebook (cheap to store; avoids re-calibrating)
    )
    if full_taps:
        for k, v in rt.items():
            if k not in ("idx", "loss"):
                arrays["tap:" + k] = v.numpy()
        arrays["x"] = x.numpy()
    save(f"model_{name}", **arrays)


# ---------------------------------------------------------------- G7: autocast (BASELINE configs #3 / #4)
def gen_model_autocast(name, batch, size, dtype, tag):
    """The reference under torch.autocast('cpu', dtype) -- what `with torch.autocast('cuda')` does to the
    encoder in the extraction script (extract_embeddings.py:124-125), here for the whole VQAE.forward."""
    spec = O.SPECS[name]
    p = O.make_params(spec, 0)
    x = O.make_patches(batch, size, 0)
    p = O.calibrate_codebook(O.make_patches(2, size, 99), p, spec)
    model = S.build_reference_model(spec, p)
    t = time.time()
    with torch.autocast("cpu", dtype=dtype):
        out, losses = model(x)
        (q,), (idx,), (loss,) = model.encoder(x)
    t_ref = time.time() - t
    (_,), (idx32,), _ = model.encoder(x)
    ot = {}
    oout, olosses = O.vqae_forward(x, p, spec, ot, dtype=dtype)
    assert torch.equal(ot["idx"], idx) and torch.equal(oout, out), "oracle autocast path deviates from the reference"
    agree32 = (idx == idx32).float().mean().item()
    print(f"  {name} {tag}: reference {t_ref:.1f}s, out dtype {out.dtype}, oracle identical; "
          f"index agreement with the fp32 reference {agree32 * 100:.2f}%")
    save(f"model_{name}_{tag}", spec=np.array(repr(spec.to_dict())), batch=batch, size=size,
         idx=idx.numpy().astype(np.uint16), idx_fp32=idx32.numpy().astype(np.uint16),
         loss=np.float32(float(loss)), out_sample=out.float()[:, :, ::16, ::16].numpy(),
         out_mean=out.float().mean(dim=(1, 2, 3)).numpy(), out_std=out.float().std(dim=(1, 2, 3)).numpy(),
         q_sample=q.float()[:, ::8, ::4, ::4].numpy(), embed=p["encoder.vq_layers.0.embed"].numpy())


# ---------------------------------------------------------------- G9: per-block taps of the mid-size models
def tap_sample(t):
    """Strided sample of an activation tap [B,C,H,W]: channels ::4 (::2 below 16), 8x8 spatial positions (phase 1 so
    interior pixels are sampled, not only tile corners)."""
    cs = 4 if t.shape[1] >= 16 else 2
    s = max(1, t.shape[2] // 8)
    return t[:, ::cs, 1::s, 1::s]


def gen_model_taps(name, batch, size, dtype, tag):
    """The reference run block by block (fp32, or under torch.autocast('cpu', dtype) as in
    extract_embeddings.py:124-125) on a model whose levels reach the production kernels; records a strided sample
    and the exact fp64 sum / sum of squares of EVERY block output.  The oracle must reproduce all of them bit for
    bit; the GPU tests then compare each HIP block against the oracle's full tensors (tests/test_blocks_gpu.py)."""
    import contextlib
    spec = O.SPECS[name]
    p = O.make_params(spec, 0)
    x = O.make_patches(batch, size, 0)
    p = O.calibrate_codebook(O.make_patches(2, size, 99), p, spec)
    model = S.build_reference_model(spec, p)
    ctx = (lambda: torch.autocast("cpu", dtype=dtype)) if dtype is not None else contextlib.nullcontext
    with ctx():
        rt = ref_forward_with_taps(model, x)
        out, _ = model(x)
    assert torch.equal(rt["out"], out)
    ot = {}
    oout, olosses = O.vqae_forward(x, p, spec, ot, dtype=dtype)
    assert torch.equal(oout, out) and torch.equal(ot["idx"], rt["idx"])
    arrays = dict(spec=np.array(repr(spec.to_dict())), batch=batch, size=size, tag=np.array(tag),
                  idx=rt["idx"].numpy().astype(np.uint16), loss=np.float32(float(rt["loss"])),
                  embed=p["encoder.vq_layers.0.embed"].numpy())
    n = 0
    for k, v in rt.items():
        if k in ("idx", "loss"):
            continue
        v = v.float()
        assert torch.equal(ot[k].float(), v) if k in ot else True, k       # oracle == reference on the FULL tensor
        arrays["tap:" + k] = tap_sample(v).numpy()
        arrays["sum:" + k] = np.array([v.double().sum().item(), (v.double() ** 2).sum().item()])
        n += 1
    print(f"  {name} {tag}: {n} taps, oracle identical on every full tensor; codes used "
          f"{rt['idx'].unique().numel()}/{spec.num_embeddings}")
    save(f"taps_{name}_{tag}", **arrays)


# ---------------------------------------------------------------- G5: driver
def gen_driver():
    S.install()
    import importlib.util
    spec_ = importlib.util.spec_from_file_location(
        "ref_extract", os.path.join(S.REFERENCE_ROOT, "scripts/extract_embeddings/extract_embeddings.py"))
    ref = importlib.util.module_from_spec(spec_)
    sys.modules["ref_extract"] = ref
    spec_.loader.exec_module(ref)

    # two slides (6x5 and 3x4 tiles of 4x4 codes), tiles interleaved across batches of 7, in the
    # dataset's row-major order (datamodules/camelyon16.py:184-190).  run_eval needs CUDA + ASAP,
    # so it is replaced by a generator with the same yield contract (extract_embeddings.py:132-138);
    # get_encodings (:43-89) -- the function under test -- runs unmodified.
    sizes = np.array([[6, 5], [3, 4]])
    names = ["images/slide_a", "images/slide_b"]
    rng = np.random.Generator(np.random.PCG64(5))
    tiles, meta = [], []
    for s, (r, c) in enumerate(sizes):
        for i in range(r * c):
            hi = 200 if s == 0 else 2            # slide_b only has codes {0,1} -> bool cast
            tiles.append(rng.integers(0, hi, size=(4, 4), dtype=np.int64))
            meta.append((s, i // c, i % c))
    tiles = np.stack(tiles)

    class DS:
        _sizes = sizes
        _lengths = sizes.prod(axis=-1)

    def fake_run_eval(model, dataset, batch_size=7):
        for b0 in range(0, len(tiles), batch_size):
            sl = slice(b0, b0 + batch_size)
            enc = torch.from_numpy(tiles[sl])
            img_idx = torch.tensor([m[0] for m in meta[sl]])
            patch_idx = torch.tensor([[m[1], m[2]] for m in meta[sl]])
            nm = [names[m[0]] for m in meta[sl]]
            yield ((enc, nm, img_idx, patch_idx),)

    ref.run_eval = fake_run_eval

    # torch 1.11 (the reference's pin) treats `tensor[ndarray of shape [2,n,h,w]]` as the index TUPLE
    # (rows, cols) (legacy treat-sequence-as-tuple rule); torch 2.10 converts the ndarray to one
    # index tensor instead and get_encodings:84 raises.  Restore the pinned version's semantics for
    # the slide-grid tensor only; the function under test is still executed unmodified.
    class LegacyIndexTensor(torch.Tensor):
        def __setitem__(self, index, value):
            if isinstance(index, np.ndarray) and index.ndim > 1:
                index = tuple(torch.from_numpy(np.ascontiguousarray(i)) for i in index)
            return super().__setitem__(index, value)

    class TorchProxy:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def empty(*a, **k):
            return torch.empty(*a, **k).as_subclass(LegacyIndexTensor)

    ref.torch = TorchProxy()
    got = dict(ref.get_encodings(None, DS()))
    arrays = {"tiles": tiles, "meta": np.array(meta), "sizes": sizes}
    for n, a in got.items():
        print(f"  driver {n}: {a.shape} {a.dtype}")
        arrays["grid:" + n] = a
        r, c = sizes[names.index(n)]
        mine = O.cast_to_lowest_dtype(O.stitch_slide(tiles[[i for i, m in enumerate(meta) if names[m[0]] == n]], r, c))
        assert mine.dtype == a.dtype and np.array_equal(mine, a)
    save("driver", **arrays)


# ---------------------------------------------------------------- G6: training-mode EMA bookkeeping
def gen_ema():
    S.install()
    from vq_ae.layers.vq import EMAVectorQuantizer
    D, K = 16, 32
    z0, embed = O.make_vq_case(D, K, 1024, seed=3, adversarial=False)
    z1, _ = O.make_vq_case(D, K, 1024, seed=4, adversarial=False)
    vq = EMAVectorQuantizer(num_embeddings=K, embedding_dim=D, commitment_cost=1.0, decay=0.99,
                            laplace_alpha=1e-5).train()
    vq.embed.copy_(embed); vq.embed_avg.copy_(embed)
    out = {}
    for step, z in enumerate((z0 * 1.7 + 0.3, z1 * 1.7 + 0.3)):
        zin = z.reshape(1, 32, 32, D).permute(0, 3, 1, 2).contiguous()
        q, idx, loss = vq(zin)
        out[f"idx{step}"] = idx.reshape(-1).numpy().astype(np.uint16)
        out[f"embed{step}"] = vq.embed.numpy().copy()
        out[f"embed_avg{step}"] = vq.embed_avg.numpy().copy()
        out[f"cluster_size{step}"] = vq.cluster_size.numpy().copy()
        out[f"loss{step}"] = np.float32(loss.item())
    # oracle restatement of _init_ema/_update_ema
    e, ea, cs = O.init_ema((z0 * 1.7 + 0.3), embed, embed.clone(), torch.zeros(K))
    for step, z in enumerate((z0 * 1.7 + 0.3, z1 * 1.7 + 0.3)):
        idx, _, _ = O.vq_argmin_p4(z, e, 4.0)
        assert np.array_equal(idx.numpy(), out[f"idx{step}"].astype(np.int64))
        e, ea, cs = O.update_ema(z, idx, ea, cs, 0.99, 1e-5)
        assert np.allclose(e.numpy(), out[f"embed{step}"], rtol=1e-6, atol=1e-7)
        assert np.allclose(cs.numpy(), out[f"cluster_size{step}"], rtol=1e-6, atol=1e-7)
    print("  ema: oracle matches reference training-mode bookkeeping")
    save("ema", D=D, K=K, **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="vq,vqnd,tiny,tinyP,B,A,C,driver,ema,autocast,mbconv,taps")
    args = ap.parse_args()
    todo = args.only.split(",")
    torch.manual_seed(0)
    if "vq" in todo:
        print("G1 vq"); gen_vq()
    if "vqnd" in todo:
        print("G1b vq on 3-D / 5-D inputs"); gen_vq_nd()
    if "tiny" in todo:
        print("G2 tiny"); gen_model("tiny", 2, 32, True)
    if "tinyP" in todo:
        print("G2 tinyP"); gen_model("tinyP", 2, 32, True)
    if "B" in todo:
        print("G3 cfg B b=4 (BASELINE config #1)"); gen_model("B", 4, 256, False)
    if "A" in todo:
        print("G4 cfg A b=2"); gen_model("A", 2, 512, False)
    if "C" in todo:
        print("G4 cfg C b=1"); gen_model("C", 1, 256, False)
    if "autocast" in todo:
        print("G7 autocast")
        gen_model_autocast("tinyP", 2, 32, torch.bfloat16, "bf16")
        gen_model_autocast("tiny", 2, 32, torch.float16, "f16")
        gen_model_autocast("B", 2, 256, torch.bfloat16, "bf16")
        gen_model_autocast("A", 2, 512, torch.bfloat16, "bf16")      # BASELINE config #3
        gen_model_autocast("C", 1, 256, torch.float16, "f16")        # BASELINE config #4
        gen_model_autocast("tinyM", 2, 32, torch.float16, "f16")     # the MBConv variant under the extraction default (conv_block.py:240-321)
        gen_model_autocast("tinyM", 2, 32, torch.bfloat16, "bf16")
    if "mbconv" in todo:
        print("G8 MBConv / EfficientNetV2 variant")
        gen_model("tinyM", 2, 32, True)
        gen_model("BM", 2, 256, False)
    if "taps" in todo:
        print("G9 per-block taps (mid-size models reaching the production kernels)")
        for nm, nb, sz in (("mid", 2, 128), ("mid16", 2, 128), ("midA", 2, 128), ("midC", 2, 128), ("midW", 1, 256)):
            gen_model_taps(nm, nb, sz, None, "f32")
            gen_model_taps(nm, nb, sz, torch.bfloat16, "bf16")
            gen_model_taps(nm, nb, sz, torch.float16, "f16")
    if "driver" in todo:
        print("G5 driver"); gen_driver()
    if "ema" in todo:
        print("G6 ema"); gen_ema()
