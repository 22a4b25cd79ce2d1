"""GPU: whole-slide driver (run_eval / get_encodings mirrors) against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_get_encodings_synthetic_slides(amd, oracle):
    from vqae_amd.extract_embeddings import SyntheticSlideDataset, cast_to_lowest_dtype, get_encodings, run_eval
    g = load_golden("model_tiny")
    spec = oracle.SPECS["tiny"]
    p = oracle.make_params(spec, 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    nat = amd.NativeVQAE(amd.SPECS["tiny"], p)
    ds = SyntheticSlideDataset([(3, 2), (2, 3)], patch_size=32, raw=True, names=["a", "b"])
    grids = dict(get_encodings(nat, ds, batch_size=5, autocast_dtype=None, num_workers=0))   # fp32 convs; batches straddle the two slides
    assert set(grids) == {"images/a", "images/b", "masks/a_mask", "masks/b_mask"}
    # oracle: encode every tile on the CPU and stitch
    for s, name in enumerate(["a", "b"]):
        rows, cols = ds._sizes[s]
        tiles, labs = [], []
        for i in range(len(ds)):
            if ds.locate(i)[0] != s:
                continue
            img, lab, _ = ds[i]
            x = oracle.normalize_u8(img.numpy()[None])
            (_,), (idx,), _ = oracle.encoder_forward(x, p, spec)
            tiles.append(idx[0].numpy())
            labs.append(oracle.adaptive_max_pool_labels(lab.numpy(), 8)[0])
        want = oracle.cast_to_lowest_dtype(oracle.stitch_slide(np.stack(tiles), rows, cols))
        got = grids[f"images/{name}"]
        assert got.shape == (rows * 8, cols * 8) and got.dtype == want.dtype
        assert np.array_equal(got, want)
        wantm = oracle.cast_to_lowest_dtype(oracle.stitch_slide(np.stack(labs), rows, cols))
        assert np.array_equal(grids[f"masks/{name}_mask"], wantm) and grids[f"masks/{name}_mask"].dtype == wantm.dtype
    # reference yield contract of run_eval: ((idx, names, img_index, patch_index), (labels, ...))
    first = next(iter(run_eval(nat, ds, batch_size=4, num_workers=0)))
    (idx, names, ii, pi), (lab, lnames, _, _) = tuple(first)
    assert idx.dtype == torch.int64 and idx.shape == (4, 8, 8) and names[0] == "images/a" and lnames[0] == "masks/a_mask"
    assert pi.shape == (4, 2) and lab.shape == (4, 8, 8)


def test_module_mirror_works_with_driver(amd, oracle):
    from vqae_amd.extract_embeddings import SyntheticSlideDataset, get_encodings
    from vqae_amd.model import VQAE
    g = load_golden("model_tiny")
    p = oracle.make_params(oracle.SPECS["tiny"], 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    model = VQAE.from_spec(amd.SPECS["tiny"])
    model.load_state_dict(p, strict=False)
    model = model.cuda().eval()
    ds = SyntheticSlideDataset([(2, 2)], patch_size=32, raw=False)
    grids = dict(get_encodings(model, ds, batch_size=3, num_workers=0))
    nat = amd.NativeVQAE(amd.SPECS["tiny"], p)
    ds_raw = SyntheticSlideDataset([(2, 2)], patch_size=32, raw=True)
    grids_raw = dict(get_encodings(nat, ds_raw, batch_size=3, num_workers=0))
    assert np.array_equal(grids["images/slide_000"], grids_raw["images/slide_000"])


def test_pinned_ring_is_really_page_locked(amd):
    """PinnedRing.pin_async registers the shared mapping slot by slot on a helper thread (round 3: an earlier form handed
    torch.cuda.set_device a bare 'cuda' device there, the thread died, and the ring silently stayed pageable): every slot ends up
    page-locked as torch sees it, for a device given with and without an index, and close() releases it."""
    import warnings
    from vqae_amd.extract_embeddings import PinnedRing
    for dev in (torch.device("cuda"), torch.device("cuda", 0), None):
        ring = PinnedRing(5, 3, (16, 16, 3), torch.uint8, (1, 16, 16), torch.uint8, pin=True, device=dev)
        with warnings.catch_warnings():
            warnings.simplefilter("error")                            # a failed registration warns: make that a failure here
            ring.pin_async()
            assert ring.wait_pinned(0) and ring.wait_pinned(4)
            assert ring.wait_pinned() is True and ring.pin_error is None
        for slot in range(5):
            imgs, labs = ring.views(slot, 3)
            assert imgs.is_pinned() and labs.is_pinned()
            imgs.fill_(slot)
            assert int(imgs.to("cuda", non_blocking=True).sum().item()) == slot * imgs.numel()
        ring.close()
        assert not ring.pinned


def test_recut_encoder_stream_gives_the_same_grids(amd, oracle):
    """run_eval calls the encoder on the next multiple of 64 tiles when batch_size is not one (encode_batch="auto"; the stream of
    loader batches is re-cut, the yields are not): slide grids identical to one encoder call per loader batch, for the default and
    for forced call sizes that straddle batch and slide boundaries -- the encoder is batch-invariant bit for bit."""
    from vqae_amd.extract_embeddings import SyntheticSlideDataset, get_encodings
    g = load_golden("model_tiny")
    p = oracle.make_params(oracle.SPECS["tiny"], 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    nat = amd.NativeVQAE(amd.SPECS["tiny"], p)
    ds = SyntheticSlideDataset([(9, 11), (5, 7), (12, 6)], patch_size=32, raw=True)      # 206 tiles
    want = dict(get_encodings(nat, ds, batch_size=10, num_workers=2, encode_batch=None))
    for eb in ("auto", 7, 64, 1000):
        got = dict(get_encodings(nat, ds, batch_size=10, num_workers=2, encode_batch=eb))
        assert sorted(got) == sorted(want)
        for k in want:
            assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), (eb, k)


def test_save_encodings_hdf5_streams_slide_grids(amd, oracle, tmp_path):
    """BASELINE config 5 shape: slides -> HIP encoder -> stitched grids -> one HDF5 file (groups images/masks,
    keys <stem> / <stem>_mask: convert.py:27-32, camelyon16.py:226-235)."""
    from vqae_amd import hdf5
    from vqae_amd.extract_embeddings import SyntheticSlideDataset, get_encodings, save_encodings, save_encodings_hdf5, convert_npy_to_hdf5
    g = load_golden("model_tiny")
    p = oracle.make_params(oracle.SPECS["tiny"], 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
    nat = amd.NativeVQAE(amd.SPECS["tiny"], p)
    ds = SyntheticSlideDataset([(3, 2), (1, 4), (2, 2)], patch_size=32, raw=True, names=["normal_001", "tumor_002", "test_003"])
    want = dict(get_encodings(nat, ds, batch_size=7, num_workers=2))
    out = save_encodings_hdf5(tmp_path / "direct.hdf5", nat, ds, batch_size=7, num_workers=2)
    r = hdf5.H5Reader(out)
    assert sorted(r.keys()) == ["images", "masks"]
    for key in ("normal_001", "tumor_002", "test_003"):
        a, m = r["images"][key], r["masks"][key + "_mask"]
        assert a.dtype == want["images/" + key].dtype and np.array_equal(a, want["images/" + key])
        assert m.dtype == want["masks/" + key + "_mask"].dtype and np.array_equal(m, want["masks/" + key + "_mask"])
    # the reference's two-step route (.npy per slide, then the converter) gives the same file content
    save_encodings(tmp_path, nat, ds, batch_size=7, num_workers=2)
    two_step = hdf5.read_hdf5(convert_npy_to_hdf5(tmp_path / "encodings"))
    direct = hdf5.read_hdf5(out)
    assert all(np.array_equal(two_step[g_][n], direct[g_][n]) and two_step[g_][n].dtype == direct[g_][n].dtype
               for g_ in direct for n in direct[g_])


def test_run_eval_default_is_the_reference_fp16_autocast(amd, oracle):
    """run_eval's defaults are the reference's (extract_embeddings.py:92-101,124-125): batch 100, 6 workers,
    prefetch 5 and the encoder under fp16 autocast.  With the defaults a NativeVQAE AND a module mirror produce the
    codes of the reference under autocast(float16) (fixture model_tiny_f16, recorded from the imported reference);
    autocast_dtype=None gives the fp32 codes (fixture model_tiny)."""
    import inspect
    from vqae_amd.extract_embeddings import run_eval
    from vqae_amd.model import VQAE
    sig = inspect.signature(run_eval).parameters
    assert sig["batch_size"].default == 100 and sig["num_workers"].default == 6 and sig["prefetch_factor"].default == 5
    assert sig["autocast_dtype"].default == torch.float16
    g16, g32 = load_golden("model_tiny_f16"), load_golden("model_tiny")
    p = oracle.make_params(oracle.SPECS["tiny"], 0)
    p["encoder.vq_layers.0.embed"] = torch.from_numpy(g16["embed"])
    x = oracle.make_patches(int(g16["batch"]), 32, 0)

    class Patches(torch.utils.data.Dataset):                      # the fixture's two patches as a one-slide dataset
        _sizes = np.array([[1, 2]]); _lengths = np.array([2]); _cum_lengths = np.array([2])
        image_paths = ["/d/images/s.tif"]; mask_paths = ["/d/masks/s_mask.tif"]; patch_size = (32, 32)

        def __len__(self):
            return 2

        def __getitem__(self, i):
            return x[i], torch.zeros(1, 32, 32, dtype=torch.uint8), (0, np.asarray((0, i)), self.image_paths[0], self.mask_paths[0])

    nat = amd.NativeVQAE(amd.SPECS["tiny"], p)
    model = VQAE.from_spec(amd.SPECS["tiny"])
    model.load_state_dict(p, strict=False)
    model = model.cuda().eval()
    for m in (nat, model):
        (idx, *_), _ = tuple(next(iter(run_eval(m, Patches()))))              # all defaults (6 workers, fp16 autocast)
        agree16 = float((idx.cpu().numpy() == g16["idx"].astype(np.int64)).mean())
        (idx32, *_), _ = tuple(next(iter(run_eval(m, Patches(), autocast_dtype=None, num_workers=0))))
        assert np.array_equal(idx32.cpu().numpy(), g32["idx"].astype(np.int64))
        assert agree16 >= 0.99, agree16
    assert nat.compute_dtype == 0 and nat.with_dtype(torch.float16).compute_dtype == 2


def _sharded_gpu_worker(rank, ws, port, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import vqae_amd
        from oracle import vqae_oracle as O
        from vqae_amd.extract_embeddings import SyntheticSlideDataset, get_encodings
        torch.cuda.set_device(0)
        g = load_golden("model_tiny")
        p = O.make_params(O.SPECS["tiny"], 0)
        p["encoder.vq_layers.0.embed"] = torch.from_numpy(g["embed"])
        nat = vqae_amd.NativeVQAE(vqae_amd.SPECS["tiny"], p)
        ds = SyntheticSlideDataset([(3, 2), (2, 3)], patch_size=32, raw=True, names=["a", "b"])
        alone = dict(get_encodings(nat, ds, batch_size=5, num_workers=0, shard=False))      # this rank encodes everything
        same = lambda got: set(got) == set(alone) and all(np.array_equal(got[k], alone[k]) and got[k].dtype == alone[k].dtype for k in alone)
        sharded = dict(get_encodings(nat, ds, batch_size=5, num_workers=0))                 # ranks split every batch 3 + 2
        # default: only rank 0 (the writer) stitches and downloads; rank 1 drives the collectives and yields (name, None)
        ok = same(sharded) if rank == 0 else (set(sharded) == set(alone) and all(v is None for v in sharded.values()))
        ok = ok and same(dict(get_encodings(nat, ds, batch_size=5, num_workers=0, replicate=True)))
        ok = ok and same(dict(get_encodings(nat, ds, batch_size=5, num_workers=2, prefetch_factor=2, loader="ring", replicate=True)))
        q.put((rank, ok, "" if ok else f"rank {rank}: sharded grids differ from the single-process grids"))
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()))
    dist.destroy_process_group()


def test_sharded_get_encodings_two_ranks_on_one_gpu(amd):
    """The sharded run_eval / get_encodings path with the real HIP encoder: two processes (one GPU here, so both use
    cuda:0 and the share gather travels over gloo through the host -- dist.all_gather_shares) must each assemble exactly
    the grids a single process produces.  On a multi-GPU node the same code runs one rank per GPU over RCCL."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 90
    procs = [ctx.Process(target=_sharded_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), [r[2] for r in res if not r[1]]
