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
