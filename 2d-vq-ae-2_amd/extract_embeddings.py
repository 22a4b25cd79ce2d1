"""Whole-slide batched driver: mirror of scripts/extract_embeddings/extract_embeddings.py
(`run_eval` :92-138, `get_encodings` :43-89, `cast_to_lowest_dtype` :54-59, `.npy` layout :183-185)
on the HIP path, plus the pieces the reference takes from elsewhere and that cannot run here
(ASAP + CAMELYON16 TIFFs are absent): a synthetic slide dataset with the index -> (slide, row, col)
contract of CAMELYON16SlicePatchDataSet (datamodules/camelyon16.py:160-211), device-side uint8
ingestion, label max-pooling, and patch-batch sharding over ranks (dist.py).

Out of scope: Hydra/Lightning checkpoint discovery (`main`, `find_ckpt_folder`).
"""
import contextlib
import os
from pathlib import Path
from typing import Iterator, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler
from torch.utils.data.dataloader import default_collate

from . import dist as vdist
from . import hdf5
from . import ops

MEAN = (0.7279, 0.5955, 0.7762)      # conf/transforms/camelyon16_transforms.yaml:15-23
STD = (0.2419, 0.3083, 0.1741)


class CheckpointNotFoundError(ValueError):       # extract_embeddings.py:35-36
    ...


class TooManyCheckpointsError(ValueError):       # extract_embeddings.py:39-40
    ...


def cast_to_lowest_dtype(array: np.ndarray) -> np.ndarray:
    """extract_embeddings.py:54-59: bool if the values are exactly {0, 1}, else the smallest integer
    dtype holding [min, max]."""
    amin, amax = array.min(), array.max()
    if amin == 0 and amax == 1:
        return array.astype(bool)
    return array.astype(np.result_type(np.min_scalar_type(amin), np.min_scalar_type(amax)))


class SyntheticSlideDataset(Dataset):
    """Non-overlapping tiles of synthetic slides with the item contract of
    CAMELYON16SlicePatchDataSet.__getitem__ (camelyon16.py:170-211):
        (patch, label, (img_index, patch_indices[row, col], image_path, mask_path))
    index -> slide via cumulative tile counts (bisect, :184), row = i // cols, col = i % cols (:187-190).
    raw=True returns uint8 HWC patches (as the WSI reader delivers, imagereader.py:473) for on-device
    normalisation; raw=False returns the normalised CHW fp32 tensor ToTensorV2 would."""

    def __init__(self, sizes, patch_size=512, seed=0, raw=True, names=None):
        self._sizes = np.asarray(sizes, dtype=np.int64).reshape(-1, 2)       # tiles (rows, cols) per slide
        self._lengths = self._sizes.prod(axis=-1)
        self._cum_lengths = np.cumsum(self._lengths)
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.seed, self.raw = seed, raw
        n = len(self._sizes)
        self.image_paths = [f"/synthetic/images/{(names[i] if names else f'slide_{i:03d}')}.tif" for i in range(n)]
        self.mask_paths = [f"/synthetic/masks/{(names[i] if names else f'slide_{i:03d}')}_mask.tif" for i in range(n)]

    def __len__(self):
        return int(self._cum_lengths[-1])

    def locate(self, index: int):
        img_index = int(np.searchsorted(self._cum_lengths, index, side="right"))
        patch_index = index - (int(self._cum_lengths[img_index - 1]) if img_index else 0)
        cols = int(self._sizes[img_index, 1])
        return img_index, patch_index // cols, patch_index % cols

    def __getitem__(self, index):
        img_index, r, c = self.locate(index)
        rng = np.random.Generator(np.random.PCG64([self.seed, img_index, r, c]))
        h, w = self.patch_size
        patch = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        label = (rng.random((h, w)) > 0.995).astype(np.uint8)                 # sparse "tumour" pixels
        if self.raw:
            img = torch.from_numpy(patch)
        else:
            mean = np.array(MEAN, np.float32) * np.float32(255)
            inv = np.float32(1) / (np.array(STD, np.float32) * np.float32(255))
            img = torch.from_numpy(((patch.astype(np.float32) - mean) * inv).transpose(2, 0, 1).copy())
        return img, torch.from_numpy(label)[None], (img_index, np.asarray((r, c)), self.image_paths[img_index],
                                                    self.mask_paths[img_index])


def _extract_path(path: str) -> str:                 # extract_embeddings.py:111-112
    return Path(path).parent.stem + '/' + Path(path).stem


_REF_AUTOCAST = torch.float16        # `with torch.autocast('cuda')` (extract_embeddings.py:124): CUDA autocast defaults to fp16


def _encode(model, imgs, autocast_dtype=_REF_AUTOCAST):
    """indices [B, h, w] from whatever was passed: NativeVQAE, or a module with `.encoder`
    (reference contract: `tuple(zip(*model.encoder(imgs)))[0]` = (q, idx, loss), :125), with the convolutions
    in `autocast_dtype` (None = fp32): a NativeVQAE switches to the handle of that compute dtype, a module mirror
    runs inside `torch.autocast('cuda', dtype)` exactly as the reference wraps its encoder (:124-125)."""
    if hasattr(model, "with_dtype"):                               # NativeVQAE
        nat = model.with_dtype(autocast_dtype)
        if imgs.dtype == torch.uint8:
            return nat.encode_u8(imgs)[1]
        return nat.encode(imgs, "NCHW", want_q=False, want_loss=False)[1]
    ctx = torch.autocast("cuda", dtype=autocast_dtype) if autocast_dtype is not None else contextlib.nullcontext()
    with ctx:
        nat = getattr(model, "native", None)
        if imgs.dtype == torch.uint8 and nat is not None:
            return nat().encode_u8(imgs)[1]
        _, idx, _ = tuple(zip(*model.encoder(imgs)))[0]
    return idx


def _factor(model) -> int:
    """Down-sampling factor 2**n_down of the encoder (patch size / code-grid size)."""
    if hasattr(model, "factor"):
        return int(model.factor)
    if hasattr(model, "native"):
        return int(model.native().factor)
    raise TypeError("run_eval: model must be a NativeVQAE or a vqae_amd.model.{VQAE,Encoder}")


def _num_codes(model) -> Optional[int]:
    spec = getattr(model, "spec", None)
    if spec is None and hasattr(model, "_spec"):
        spec = model._spec()
    return int(spec.num_embeddings) if spec is not None else None


def batch_meta(dataset, lo: int, hi: int):
    """(img_index [n] int64, patch_index [n, 2] int64, image paths, mask paths) of items [lo, hi) WITHOUT reading a
    pixel: the index -> (slide, row, col) rule of CAMELYON16SlicePatchDataSet.__getitem__ (camelyon16.py:184-190:
    bisect over `_cum_lengths`, row = i // cols, col = i % cols), which the default collate would otherwise deliver."""
    idx = np.arange(lo, hi, dtype=np.int64)
    cum = np.asarray(dataset._cum_lengths, dtype=np.int64)
    sizes = np.asarray(dataset._sizes, dtype=np.int64)
    img = np.searchsorted(cum, idx, side="right")                        # bisect.bisect
    first = np.where(img > 0, cum[np.maximum(img - 1, 0)], 0)
    local = idx - first
    cols = sizes[img, 1]
    patch = np.stack([local // cols, local % cols], 1)
    ip, mp = np.asarray(dataset.image_paths), np.asarray(dataset.mask_paths)
    return torch.from_numpy(img), torch.from_numpy(patch), [str(v) for v in ip[img]], [str(v) for v in mp[img]]


class ShardBatchSampler(Sampler):
    """Batch sampler of one rank: of every global batch [k * bs, (k + 1) * bs) it yields only this rank's contiguous
    share (dist.shard_range), so a rank's loader workers read only the tiles that rank encodes -- host loading is
    divided over the ranks, not replicated.  Every rank walks the same number of batches (a share may be empty)."""

    def __init__(self, n, batch_size, rank, world_size):
        self.n, self.bs, self.rank, self.ws = int(n), int(batch_size), rank, world_size

    def __len__(self):
        return -(-self.n // self.bs)

    def __iter__(self):
        for b0 in range(0, self.n, self.bs):
            lo, hi = vdist.shard_range(min(self.bs, self.n - b0), self.rank, self.ws)
            yield list(range(b0 + lo, b0 + hi))


def _collate(batch):
    return default_collate(batch) if batch else None


def _pool_labels(lab, out_hw):
    """adaptive_max_pool2d(labels, encoding grid) (extract_embeddings.py:127-130) on the device."""
    return ops.label_maxpool(lab.reshape(lab.shape[0], lab.shape[-2], lab.shape[-1]).to(torch.uint8), out_hw)


@torch.no_grad()
def run_eval(model, dataset, batch_size=100, *, autocast_dtype=_REF_AUTOCAST, num_workers=6, prefetch_factor=5,
             device=None, shard=True, encode_fn=None, pool_fn=None):
    """Batched encoder pass: drop-in for run_eval (extract_embeddings.py:92-138).  Yields, per batch, the reference's pair
        ((encoding_indices, names, img_index, patch_index), (labels_pooled, names, img_index, patch_index))
    with tensors on `device`.  Defaults are the reference's: batch 100, 6 loader workers, prefetch 5, pinned memory
    (:95-101), the encoder under fp16 autocast (:124-125; pass autocast_dtype=None for fp32 convolutions, or
    torch.bfloat16).

    Under torch.distributed (one process per GPU) every global batch is sharded contiguously over the ranks: a
    rank's loader reads and its GPU encodes only its share (ShardBatchSampler); the shares are re-assembled with ONE
    fixed-size all-gather per batch (dist.all_gather_shares: uint8 / uint16 code tiles + pooled label tiles in one
    buffer, no size exchange, no host sync), issued asynchronously so it overlaps the next batch's encoder pass.
    Every rank yields the full batch in the original order.

    encode_fn(imgs_on_device) -> indices [b, h, w] and pool_fn(labels_on_device, out_hw) -> [b, out_hw, out_hw]
    replace the HIP encoder / max-pool (used by the CPU tests of the sharded path; the product default has no
    CPU fallback and raises without a GPU)."""
    device = torch.device(device) if device is not None else torch.device("cuda")
    rank, ws = vdist.world() if shard else (0, 1)
    enc = encode_fn or (lambda x: _encode(model, x, autocast_dtype))
    pool = pool_fn or _pool_labels
    pin = device.type == "cuda"
    extra = {"prefetch_factor": prefetch_factor} if num_workers else {}
    if ws > 1:
        loader = DataLoader(dataset, batch_sampler=ShardBatchSampler(len(dataset), batch_size, rank, ws),
                            collate_fn=_collate, pin_memory=pin, num_workers=num_workers, **extra)
    else:
        loader = DataLoader(dataset, batch_size=batch_size, pin_memory=pin, num_workers=num_workers, **extra)
    factor = _factor(model) if encode_fn is None else None
    n_codes = _num_codes(model) if model is not None else None
    code_bytes = 1 if (n_codes or 1 << 30) <= 256 else (2 if (n_codes or 1 << 30) <= 65536 else 4)
    code_dtype = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[code_bytes]
    cap = vdist.gather_capacity(batch_size, ws)

    def finish(item):
        """gathered buffer -> the reference's generator of two tuples, for one batch"""
        (out, work, n, th, tw, meta, labels_dtype) = item
        if work is not None:
            work.wait()
        idx_parts, pool_parts = [], []
        nb = th * tw * code_bytes
        for r in range(ws):
            lo, hi = vdist.shard_range(n, r, ws)
            rows = out[r, : hi - lo]
            idx_parts.append(rows[:, :nb].contiguous().view(code_dtype).reshape(hi - lo, th, tw))
            pool_parts.append(rows[:, nb:].reshape(hi - lo, th, tw))
        idx = torch.cat(idx_parts, 0).to(torch.int64)
        if code_bytes == 2:
            idx = idx & 0xFFFF                                     # uint16 codes travelled as int16 bit patterns
        pooled = torch.cat(pool_parts, 0).to(labels_dtype)
        img_index, patch_index, img_path, label_path = meta
        return ((data, list(map(_extract_path, paths)), img_index, patch_index)
                for data, paths in ((idx, img_path), (pooled, label_path)))

    pending, grid_hw = None, None
    for k, batch in enumerate(loader):
        if ws == 1:
            imgs, labels, (img_index, patch_index, img_path, label_path) = batch
            x = imgs.to(device, non_blocking=True)
            lab = labels.to(device, non_blocking=True)
            idx = enc(x)
            pooled = pool(lab, idx.shape[-1]).to(labels.dtype)
            yield ((data, list(map(_extract_path, paths)), img_index, patch_index)
                   for data, paths in ((idx, img_path), (pooled.reshape(idx.shape), label_path)))
            continue
        # ---- sharded: encode my share, pack, launch the gather, and only then hand out the PREVIOUS batch ----------
        b0 = k * batch_size
        n = min(batch_size, len(dataset) - b0)
        meta = batch_meta(dataset, b0, b0 + n)
        if batch is not None:
            imgs, labels, _ = batch
            x = imgs.to(device, non_blocking=True)
            lab = labels.to(device, non_blocking=True)
            idx = enc(x)
            th, tw = int(idx.shape[-2]), int(idx.shape[-1])
            pooled = pool(lab, tw).reshape(idx.shape[0], th * tw).to(torch.uint8)
            codes = idx.to(code_dtype).reshape(idx.shape[0], th * tw).contiguous().view(torch.uint8)
            mine = torch.zeros((cap, codes.shape[1] + pooled.shape[1]), dtype=torch.uint8, device=device)
            mine[: idx.shape[0], : codes.shape[1]] = codes
            mine[: idx.shape[0], codes.shape[1]:] = pooled
            labels_dtype = labels.dtype
            grid_hw = (th, tw, labels_dtype)
        else:                                                      # empty share (a last batch shorter than the world size)
            if grid_hw is None:                                    # ... before this rank ever encoded a tile
                ps = getattr(dataset, "patch_size", None)
                assert factor is not None and ps is not None, "run_eval: an empty share needs dataset.patch_size and a model factor"
                grid_hw = (int(ps[0]) // factor, int(ps[1]) // factor, torch.uint8)
            th, tw, labels_dtype = grid_hw
            mine = torch.zeros((cap, th * tw * (code_bytes + 1)), dtype=torch.uint8, device=device)
        out, work = vdist.all_gather_shares(mine, async_op=True)
        if pending is not None:
            yield finish(pending)
        pending = (out, work, n, th, tw, meta, labels_dtype)
    if pending is not None:
        yield finish(pending)


def _stitch(sel, rc, grid):
    return ops.stitch_tiles(sel, rc, grid)


def get_encodings(model, dataset, batch_size=100, stitch_fn=None, **kw) -> Iterator[Tuple[str, np.ndarray]]:
    """Stitch code tiles into one `[32*rows, 32*cols]` grid per slide on the device and yield
    `(name, ndarray)` -- cast to the lowest dtype -- as soon as every tile of a slide has been seen
    (extract_embeddings.py:43-89).  `stitch_fn(tiles, rc, grid)` replaces the HIP scatter (CPU tests only)."""
    stitch = stitch_fn or _stitch
    arrays, counts = {}, {}
    for ret_values in run_eval(model, dataset, batch_size=batch_size, **kw):
        for (encodings, names, img_idx, patch_idx) in ret_values:
            th, tw = int(encodings.shape[1]), int(encodings.shape[2])
            names = np.asarray(names)
            u_names, u_idx, u_counts = np.unique(names, return_counts=True, return_index=True)
            img_idx_np = np.asarray(img_idx)
            for name, image_index, count in zip(u_names, img_idx_np[u_idx], u_counts):
                if name not in counts:
                    counts[name] = int(dataset._lengths[image_index])
                    r, c = (int(v) for v in dataset._sizes[image_index])
                    arrays[name] = torch.empty((r * th, c * tw), dtype=encodings.dtype, device=encodings.device)
                mask = torch.as_tensor(img_idx_np == image_index)
                sel = encodings[mask.to(encodings.device)]
                rc = torch.as_tensor(np.asarray(patch_idx)[mask.numpy()], dtype=torch.int32, device=encodings.device)
                stitch(sel, rc, arrays[name])                              # device scatter (:83-84)
                counts[name] -= int(count)
                if counts[name] == 0:
                    counts.pop(name)
                    yield str(name), cast_to_lowest_dtype(arrays.pop(name).cpu().numpy())


def save_encodings(root, model, dataset, **kw):
    """`np.save(<root>/encodings/<images|masks>/<stem>.npy)` per slide (extract_embeddings.py:179-185).
    Under torch.distributed only rank 0 writes."""
    rank, _ = vdist.world()
    written = []
    for name, array in get_encodings(model, dataset, **kw):
        if rank == 0:
            out = Path(root) / "encodings" / (name + ".npy")
            out.parent.mkdir(parents=True, exist_ok=True)
            np.save(str(out), array)
            written.append(str(out))
    return written


def save_encodings_hdf5(out_path, model, dataset, **kw):
    """BASELINE config 5 ("embeddings streamed to HDF5"): the slide grids of `get_encodings` go straight
    into one HDF5 file -- group `images` / `masks`, dataset `<stem>` / `<stem>_mask` (the layout
    convert.py:27-32 produces and datamodules/camelyon16.py:226-235 reads) -- without the `.npy` detour.
    Every finished slide is appended at once; under torch.distributed only rank 0 writes."""
    rank, _ = vdist.world()
    writer = hdf5.H5Writer(out_path) if rank == 0 else None
    try:
        for name, array in get_encodings(model, dataset, **kw):
            if writer is not None:
                group, _, stem = name.rpartition("/")
                writer.create_dataset(group or "images", stem, array)
    finally:
        if writer is not None:
            writer.close()
    return str(out_path)


def convert_npy_to_hdf5(encodings_root, out_path=None):
    """scripts/convert_npy_embeddings_to_hdf5/convert.py:27-32: one HDF5 group per sub-directory
    (`images`, `masks`), one dataset per `.npy` stem.  Written by this package's own writer (hdf5.py;
    h5py is not installed here); h5py / libhdf5 read the result."""
    return hdf5.convert_npy_to_hdf5(encodings_root, out_path)
