"""Whole-slide batched driver: mirror of scripts/extract_embeddings/extract_embeddings.py
(`run_eval` :92-138, `get_encodings` :43-89, `cast_to_lowest_dtype` :54-59, `.npy` layout :183-185)
on the HIP path, plus the pieces the reference takes from elsewhere and that cannot run here
(ASAP + CAMELYON16 TIFFs are absent): a synthetic slide dataset with the index -> (slide, row, col)
contract of CAMELYON16SlicePatchDataSet (datamodules/camelyon16.py:160-211), device-side uint8
ingestion, label max-pooling, and patch-batch sharding over ranks (dist.py).

Out of scope: Hydra/Lightning checkpoint discovery (`main`, `find_ckpt_folder`).
"""
import os
from pathlib import Path
from typing import Iterator, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

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


def _encode(model, imgs):
    """indices [B, h, w] from whatever was passed: NativeVQAE, or a module with `.encoder`
    (reference contract: `tuple(zip(*model.encoder(imgs)))[0]` = (q, idx, loss), :125)."""
    if hasattr(model, "encode_u8") and imgs.dtype == torch.uint8:
        return model.encode_u8(imgs)[1]
    if hasattr(model, "encode"):
        return model.encode(imgs, "NCHW", want_q=False, want_loss=False)[1]
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


@torch.no_grad()
def run_eval(model, dataset, batch_size=100, num_workers=0, device=None, shard=True):
    """Batched encoder pass (extract_embeddings.py:92-138).  Yields, per batch, the reference's pair
        ((encoding_indices, names, img_index, patch_index), (labels_pooled, names, img_index, patch_index))
    with tensors on `device`.  Under torch.distributed each rank encodes a contiguous share of every
    batch and the code tiles are re-assembled with one all-gather (dist.all_gather_codes), so every
    rank yields the full batch in the original order."""
    device = device or torch.device("cuda")
    rank, ws = vdist.world() if shard else (0, 1)
    loader = DataLoader(dataset, batch_size=batch_size, pin_memory=True, num_workers=num_workers,
                        **({"prefetch_factor": 5} if num_workers else {}))
    factor = _factor(model)
    for imgs, labels, (img_index, patch_index, img_path, label_path) in loader:
        n = imgs.shape[0]
        lo, hi = vdist.shard_range(n, rank, ws)
        x = imgs[lo:hi].to(device, non_blocking=True)
        lab = labels[lo:hi].to(device, non_blocking=True)
        idx = _encode(model, x) if hi > lo else None
        out_hw = (imgs.shape[1] if imgs.dtype == torch.uint8 else imgs.shape[2]) // factor
        if idx is None:
            idx = torch.empty((0, out_hw, out_hw), dtype=torch.int64, device=device)
        pooled = ops.label_maxpool(lab.reshape(hi - lo, lab.shape[-2], lab.shape[-1]).to(torch.uint8), out_hw) \
            if hi > lo else torch.empty((0, out_hw, out_hw), dtype=torch.uint8, device=device)
        if ws > 1:
            meta = torch.stack([img_index[lo:hi].to(torch.int64), patch_index[lo:hi, 0].to(torch.int64),
                                patch_index[lo:hi, 1].to(torch.int64)], 1).to(device)
            compact = idx.to(torch.int32)
            idx, _ = vdist.all_gather_codes(compact, meta)
            idx = idx.to(torch.int64)
            pooled, _ = vdist.all_gather_codes(pooled, meta)
        yield (
            (data, list(map(_extract_path, paths)), img_index, patch_index)
            for data, paths in ((idx, img_path), (pooled.to(labels.dtype), label_path))
        )


def get_encodings(model, dataset, batch_size=100, **kw) -> Iterator[Tuple[str, np.ndarray]]:
    """Stitch code tiles into one `[32*rows, 32*cols]` grid per slide on the device and yield
    `(name, ndarray)` -- cast to the lowest dtype -- as soon as every tile of a slide has been seen
    (extract_embeddings.py:43-89)."""
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
                ops.stitch_tiles(sel, rc, arrays[name])                    # device scatter (:83-84)
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
