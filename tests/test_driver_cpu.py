"""CPU: host logic of the whole-slide driver and of the patch-batch sharding (gloo, world_size 2)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def test_cast_to_lowest_dtype_matches_reference_fixture(amd):
    from vqae_amd.extract_embeddings import cast_to_lowest_dtype
    g = load_golden("driver")
    for name in ("images/slide_a", "images/slide_b"):
        ref = g["grid:" + name]
        got = cast_to_lowest_dtype(ref.astype(np.int64))
        assert got.dtype == ref.dtype and np.array_equal(got, ref)
    assert cast_to_lowest_dtype(np.array([0, 300])).dtype == np.uint16
    assert cast_to_lowest_dtype(np.array([0, 255])).dtype == np.uint8
    assert cast_to_lowest_dtype(np.array([0, 1])).dtype == np.bool_
    assert cast_to_lowest_dtype(np.array([-1, 1])).dtype == np.int16      # result_type(int8, uint8), as the reference


def test_synthetic_slide_dataset_index_contract(amd):
    """index -> (slide, row, col) exactly as CAMELYON16SlicePatchDataSet (camelyon16.py:184-190)."""
    from vqae_amd.extract_embeddings import SyntheticSlideDataset
    ds = SyntheticSlideDataset([(6, 5), (3, 4)], patch_size=32, raw=True)
    assert len(ds) == 42 and ds._lengths.tolist() == [30, 12]
    import bisect
    cum = np.cumsum(ds._lengths)
    for index in (0, 4, 5, 29, 30, 41):
        img = bisect.bisect(cum, index)
        pi = index - (cum[img - 1] if img else 0)
        want = (img, pi // ds._sizes[img, 1], pi % ds._sizes[img, 1])
        assert ds.locate(index) == tuple(int(v) for v in want)
    img, label, (ii, rc, ip, mp_) = ds[31]
    assert img.dtype == torch.uint8 and img.shape == (32, 32, 3) and label.shape == (1, 32, 32)
    assert ii == 1 and rc.tolist() == [0, 1] and ip.endswith("slide_001.tif")
    a, _, _ = SyntheticSlideDataset([(6, 5), (3, 4)], patch_size=32, raw=True)[31]
    assert torch.equal(a, img)                                   # deterministic
    f, _, _ = SyntheticSlideDataset([(6, 5), (3, 4)], patch_size=32, raw=False)[31]
    assert f.shape == (3, 32, 32) and f.dtype == torch.float32


def test_shard_range_partitions(amd):
    from vqae_amd.dist import shard_range
    for n in (0, 1, 7, 100, 256):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _gloo_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    import vqae_amd
    from vqae_amd.dist import all_gather_codes, shard_range
    # a batch of 7 tiles (ragged over 2 ranks: 4 + 3), each rank holds its contiguous share
    tiles = torch.arange(7 * 4 * 4, dtype=torch.int32).reshape(7, 4, 4)
    meta = torch.stack([torch.arange(7) // 5, torch.arange(7) % 5, torch.arange(7) % 3], 1)
    lo, hi = shard_range(7, rank, ws)
    got, gmeta = all_gather_codes(tiles[lo:hi].clone(), meta[lo:hi].clone())
    q.put((rank, torch.equal(got, tiles) and torch.equal(gmeta, meta), lo, hi))
    dist.destroy_process_group()


def test_all_gather_codes_gloo_world2(amd):
    """The path's only collective, on CPU with gloo and 2 processes: code tiles come back in the
    original patch order on every rank, including a ragged last batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res)
    assert {(r[2], r[3]) for r in res} == {(0, 4), (4, 7)}


# ---------------------------------------------------------------------------------------------------------------
# The sharded driver end to end under gloo, world_size 2: run_eval -> get_encodings on two interleaved slides against
# the reference's own get_encodings output (tests/golden/driver.npz).  The HIP encoder / max-pool / scatter are
# replaced by table look-ups through the injection points (encode_fn, pool_fn, stitch_fn): this exercises the
# sampler, the packing, the single gather per batch, the pipelining by one batch and the slide assembly, not kernels.
class _FixtureSlides(torch.utils.data.Dataset):
    """Items carry their own index; geometry as CAMELYON16SlicePatchDataSet (_sizes, _lengths, _cum_lengths, paths)."""

    def __init__(self, sizes, label_hw=8):
        self._sizes = np.asarray(sizes, dtype=np.int64)
        self._lengths = self._sizes.prod(axis=-1)
        self._cum_lengths = np.cumsum(self._lengths)
        self.image_paths = ["/data/images/slide_a.tif", "/data/images/slide_b.tif"]
        self.mask_paths = ["/data/masks/slide_a_mask.tif", "/data/masks/slide_b_mask.tif"]
        self.patch_size = (label_hw, label_hw)
        self.reads = []

    def __len__(self):
        return int(self._cum_lengths[-1])

    def labels(self, i):
        rng = np.random.Generator(np.random.PCG64(1000 + i))
        return (rng.random((1, *self.patch_size)) > 0.8).astype(np.uint8)

    def __getitem__(self, i):
        self.reads.append(int(i))
        img = int(np.searchsorted(self._cum_lengths, i, side="right"))
        li = i - (int(self._cum_lengths[img - 1]) if img else 0)
        cols = int(self._sizes[img, 1])
        return (torch.tensor([float(i)]), torch.from_numpy(self.labels(i)),
                (img, np.asarray((li // cols, li % cols)), self.image_paths[img], self.mask_paths[img]))


def _cpu_stitch(tiles, rc, grid):
    th, tw = tiles.shape[1:]
    for t, (r, c) in zip(tiles, rc.tolist()):
        grid[r * th:(r + 1) * th, c * tw:(c + 1) * tw] = t.to(grid.dtype)
    return grid


STITCH_CALLS = []


def _counting_stitch(tiles, rc, grid):
    STITCH_CALLS.append(int(tiles.shape[0]))
    return _cpu_stitch(tiles, rc, grid)


def _run_driver(ws_label, **kw):
    from vqae_amd.extract_embeddings import get_encodings
    g = load_golden("driver")
    tiles = torch.from_numpy(g["tiles"])
    ds = _FixtureSlides(g["sizes"])
    encode = lambda x: tiles[x.reshape(-1).long()]
    pool = lambda lab, out: torch.nn.functional.adaptive_max_pool2d(lab.float().reshape(lab.shape[0], 1, *lab.shape[-2:]), out)[:, 0].to(torch.uint8)
    del STITCH_CALLS[:]
    kw.setdefault("num_workers", 0)
    got = dict(get_encodings(None, ds, batch_size=7, stitch_fn=_counting_stitch, device="cpu",
                             encode_fn=encode, pool_fn=pool, **kw))
    return g, ds, got


def _check_driver(g, ds, got, oracle):
    for name in ("images/slide_a", "images/slide_b"):
        ref = g["grid:" + name]
        assert got[name].dtype == ref.dtype and np.array_equal(got[name], ref), name
    # masks: adaptive max-pool of every tile's label, stitched (oracle restatement of extract_embeddings.py:127-130,77-84)
    first = 0
    for s, name in enumerate(("masks/slide_a_mask", "masks/slide_b_mask")):
        r, c = (int(v) for v in ds._sizes[s])
        pooled = np.stack([oracle.adaptive_max_pool_labels(ds.labels(first + i)[0][None], 4)[0] for i in range(r * c)])
        want = oracle.cast_to_lowest_dtype(oracle.stitch_slide(pooled.astype(np.int64), r, c))
        assert got[name].dtype == want.dtype and np.array_equal(got[name], want), name
        first += r * c


def test_run_eval_get_encodings_single_process(amd, oracle):
    g, ds, got = _run_driver(1)
    _check_driver(g, ds, got, oracle)
    assert sorted(ds.reads) == list(range(42))


@pytest.mark.parametrize("eb", [5, 16, 100])
def test_run_eval_recuts_the_tile_stream_for_the_encoder(amd, oracle, eb):
    """encode_batch: the encoder sees the stream of tiles in calls of `eb` tiles (the first loader batch on its own, the remainder at
    the end), the yields keep the loader's batches -- same grids as one encoder call per batch (run_eval's "auto" picks 256 when
    batch_size is not a multiple of 64: the reference's default of 100 leaves the last round of every trunk launch partly empty)."""
    from vqae_amd.extract_embeddings import get_encodings
    g = load_golden("driver")
    tiles = torch.from_numpy(g["tiles"])
    ds = _FixtureSlides(g["sizes"])
    calls = []

    def encode(x):
        calls.append(int(x.shape[0]))
        return tiles[x.reshape(-1).long()]

    pool = lambda lab, out: torch.nn.functional.adaptive_max_pool2d(lab.float().reshape(lab.shape[0], 1, *lab.shape[-2:]), out)[:, 0].to(torch.uint8)
    got = dict(get_encodings(None, ds, batch_size=7, stitch_fn=_cpu_stitch, device="cpu", encode_fn=encode, pool_fn=pool,
                             num_workers=0, encode_batch=eb))
    _check_driver(g, ds, got, oracle)
    rest = 42 - 7
    assert calls == [7] + [eb] * (rest // eb) + ([rest % eb] if rest % eb else []), calls


def test_ring_loader_equals_stock_loader(amd, oracle):
    """loader='ring': worker processes collate straight into the shared ring (PinnedRing; page-locking needs a GPU and is
    skipped here) -- same grids as the stock DataLoader path, with more batches than ring slots so that slots are reused."""
    g, ds, got = _run_driver(1, loader="ring", num_workers=2, prefetch_factor=1)      # 6 batches through 4 slots
    _check_driver(g, ds, got, oracle)
    g, ds, got = _run_driver(1, loader="torch", num_workers=2, prefetch_factor=1)
    _check_driver(g, ds, got, oracle)


def test_run_eval_reference_contract_dtypes(amd):
    """run_eval's own contract (not compact): int64 code tiles, labels in the dataset's label dtype, int64 index tensors,
    '<parent>/<stem>' names (extract_embeddings.py:111-138) -- with both loaders."""
    from vqae_amd.extract_embeddings import run_eval
    g = load_golden("driver")
    tiles = torch.from_numpy(g["tiles"])
    pool = lambda lab, out: torch.nn.functional.adaptive_max_pool2d(lab.float().reshape(lab.shape[0], 1, *lab.shape[-2:]), out)[:, 0].to(torch.uint8)
    for loader, nw in (("torch", 0), ("ring", 1)):
        ds = _FixtureSlides(g["sizes"])
        n = 0
        for (idx, names, img_index, patch_index), (pooled, mnames, _, _) in run_eval(
                None, ds, batch_size=7, device="cpu", num_workers=nw, loader=loader,
                encode_fn=lambda x: tiles[x.reshape(-1).long()], pool_fn=pool):
            assert idx.dtype == torch.int64 and pooled.dtype == torch.uint8 and pooled.shape == idx.shape
            assert img_index.dtype == torch.int64 and tuple(patch_index.shape) == (idx.shape[0], 2)
            assert torch.equal(idx, tiles[n:n + idx.shape[0]].long())
            assert all(a.startswith("images/slide_") for a in names) and all(a.startswith("masks/slide_") for a in mnames)
            n += idx.shape[0]
        assert n == 42


def test_run_eval_dataset_without_geometry_attributes(amd, oracle):
    """A wrapper dataset without _cum_lengths / _sizes / paths: the metadata comes from the collated items instead
    (ws = 1 here; under torch.distributed it travels by an object gather)."""
    from vqae_amd.extract_embeddings import run_eval
    g = load_golden("driver")
    tiles = torch.from_numpy(g["tiles"])
    inner = _FixtureSlides(g["sizes"])

    class Wrapped(torch.utils.data.Dataset):
        def __len__(self):
            return len(inner)

        def __getitem__(self, i):
            return inner[i]

    pool = lambda lab, out: torch.nn.functional.adaptive_max_pool2d(lab.float().reshape(lab.shape[0], 1, *lab.shape[-2:]), out)[:, 0].to(torch.uint8)
    for loader, nw in (("torch", 0), ("ring", 1)):
        seen = []
        for (idx, names, img_index, patch_index), _ in run_eval(None, Wrapped(), batch_size=7, device="cpu", num_workers=nw,
                                                                loader=loader, encode_fn=lambda x: tiles[x.reshape(-1).long()],
                                                                pool_fn=pool):
            seen += [(int(a), int(r), int(c)) for a, (r, c) in zip(img_index, patch_index.tolist())]
        want = []
        for s_, (r, c) in enumerate(g["sizes"]):
            want += [(s_, i // int(c), i % int(c)) for i in range(int(r) * int(c))]
        assert seen == want


def _sharded_driver_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        import vqae_amd  # noqa: F401
        from oracle import vqae_oracle
        # default: rank 0 (the only writer) stitches and downloads; rank 1 drives the same collectives and gets (name, None)
        g, ds, got = _run_driver(ws)
        names = ["images/slide_a", "images/slide_b", "masks/slide_a_mask", "masks/slide_b_mask"]
        assert sorted(got) == sorted(names), sorted(got)
        if rank == 0:
            _check_driver(g, ds, got, vqae_oracle)
            assert sum(STITCH_CALLS) == 2 * 42                      # every tile of both streams (codes, masks), once
        else:
            assert all(v is None for v in got.values()) and STITCH_CALLS == []      # no stitch, no D2H on rank 1
        reads = sorted(ds.reads)
        # replicate=True: a full copy on every rank
        g, ds, got = _run_driver(ws, replicate=True)
        _check_driver(g, ds, got, vqae_oracle)
        # ... and through the ring loader (one worker process per rank)
        g, ds, got = _run_driver(ws, replicate=True, loader="ring", num_workers=1, prefetch_factor=2)
        _check_driver(g, ds, got, vqae_oracle)
        # ... and with the encoder stream re-cut (shares of 3 / 4 tiles per batch -> encoder calls of 5: a batch's gather is launched
        # a few iterations after it was read, at different iterations on the two ranks; the collectives pair up by order)
        for eb in (5, 2, 64):
            g, ds, got = _run_driver(ws, replicate=True, encode_batch=eb)
            _check_driver(g, ds, got, vqae_oracle)
        q.put((rank, True, reads))
    except Exception as e:                                          # surface the failure in the parent
        import traceback
        q.put((rank, False, traceback.format_exc()))
    dist.destroy_process_group()


def test_sharded_run_eval_get_encodings_gloo_world2(amd):
    """Two ranks, batches of 7 over 42 tiles of two interleaved slides: every rank assembles the reference's grids
    (images and masks, incl. the bool cast), and each rank's loader READ only its own share of every batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_sharded_driver_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), [r[2] for r in res if not r[1]]
    from vqae_amd.dist import shard_range
    want = [[], []]
    for b0 in range(0, 42, 7):
        for r in range(2):
            lo, hi = shard_range(7, r, 2)
            want[r] += list(range(b0 + lo, b0 + hi))
    assert res[0][2] == want[0] and res[1][2] == want[1]           # host loading is divided, not replicated
    assert sorted(res[0][2] + res[1][2]) == list(range(42))


def _ema_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from vqae_amd.layers.vq import fused_all_reduce_stats
        g = torch.Generator().manual_seed(7)
        counts = torch.randint(0, 50, (2, 32), generator=g).float()
        dw = torch.randn(2, 32, 16, generator=g)
        c, d = fused_all_reduce_stats(counts[rank].clone(), dw[rank].clone())
        # reference: two separate all_reduce(SUM) (vq.py:57-58)
        c2, d2 = counts[rank].clone(), dw[rank].clone()
        dist.all_reduce(c2); dist.all_reduce(d2)
        ok = bool(torch.equal(c, c2) and torch.equal(d, d2))
        # the `ema` fixture (two training-mode steps of the reference, single process) with the rows of step 0 split
        # over the two ranks: per-rank code statistics -> fused all-reduce -> EMA / Laplace step (vq.py:60-74)
        from oracle import vqae_oracle as O
        gf = load_golden("ema")
        D, K = int(gf["D"]), int(gf["K"])
        z0, embed = O.make_vq_case(D, K, 1024, seed=3, adversarial=False)
        z = z0 * 1.7 + 0.3
        e, ea, cs = O.init_ema(z, embed, embed.clone(), torch.zeros(K))
        idx = torch.from_numpy(gf["idx0"].astype(np.int64))
        lo, hi = (0, 512) if rank == 0 else (512, 1024)
        n_k = torch.zeros(K).index_add_(0, idx[lo:hi], torch.ones(hi - lo))
        dw_k = torch.zeros(K, D).index_add_(0, idx[lo:hi], z[lo:hi])
        n_k, dw_k = fused_all_reduce_stats(n_k, dw_k)
        cs = cs * 0.99 + (1 - 0.99) * n_k
        ea = ea * 0.99 + (1 - 0.99) * dw_k
        tot = cs.sum()
        e_new = ea / ((cs + 1e-5) / (tot + K * 1e-5) * tot).unsqueeze(1)
        ok = ok and np.allclose(e_new.numpy(), gf["embed0"], rtol=2e-6, atol=1e-6) and \
            np.allclose(cs.numpy(), gf["cluster_size0"], rtol=2e-6, atol=1e-6)
        q.put((rank, ok, float((c - counts.sum(0)).abs().max())))
    except Exception:
        import traceback
        q.put((rank, False, traceback.format_exc()))
    dist.destroy_process_group()


def test_ema_fused_all_reduce_equals_reference_two_reduces_gloo_world2(amd):
    """_update_ema's two all_reduce(SUM) calls ([K] and [K, D], vq.py:57-58) travel as ONE flat buffer in the mirror:
    element-wise sums are unaffected by the fusion (same per-element reduction), checked under 2 ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_ema_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res
