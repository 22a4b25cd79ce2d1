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


def _has_geometry(dataset) -> bool:
    return all(hasattr(dataset, a) for a in ("_cum_lengths", "_sizes", "image_paths", "mask_paths"))


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
    divided over the ranks, not replicated.  Every rank walks the same number of batches (a share may be empty).
    `tag_batches`: the share is prefixed with -(k + 1), the batch number the ring loader's workers pick their slot by."""

    def __init__(self, n, batch_size, rank, world_size, tag_batches=False):
        self.n, self.bs, self.rank, self.ws, self.tag = int(n), int(batch_size), rank, world_size, tag_batches
        self.skip = 0                                    # batches to leave out at the front (ring loader: its probe batch)

    def __len__(self):
        return -(-self.n // self.bs) - self.skip

    def __iter__(self):
        for k, b0 in enumerate(range(0, self.n, self.bs)):
            if k < self.skip:
                continue
            lo, hi = vdist.shard_range(min(self.bs, self.n - b0), self.rank, self.ws)
            share = list(range(b0 + lo, b0 + hi))
            yield ([-(k + 1)] + share) if self.tag else share


def _collate(batch):
    return default_collate(batch) if batch else None


class StageTimer:
    """Stage breakdown of the whole-slide pipeline (tools/bench_slide.py): `host(name)` accumulates wall time of a
    host-side stage, `gpu(name)` brackets device work with events on the current stream (resolved once, in `summary`,
    so timing adds no synchronisation to the loop)."""

    def __init__(self, device=None):
        self.wall, self.events, self.device = {}, [], device
        self.cuda = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")

    @contextlib.contextmanager
    def host(self, name):
        import time
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.wall[name] = self.wall.get(name, 0.0) + time.perf_counter() - t0

    @contextlib.contextmanager
    def gpu(self, name):
        if not self.cuda:
            with self.host(name):
                yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        try:
            yield
        finally:
            b.record()
            self.events.append((name, a, b))

    def summary(self):
        out = {"host_s": {k: round(v, 4) for k, v in self.wall.items()}, "gpu_s": {}}
        if self.events:
            torch.cuda.synchronize()
            for name, a, b in self.events:
                out["gpu_s"][name] = out["gpu_s"].get(name, 0.0) + a.elapsed_time(b) * 1e-3
            out["gpu_s"] = {k: round(v, 4) for k, v in out["gpu_s"].items()}
        return out


@contextlib.contextmanager
def _stage(timer, kind, name):
    if timer is None:
        yield
    else:
        with (timer.gpu(name) if kind == "gpu" else timer.host(name)):
            yield


class PinnedRing:
    """Zero-copy host path of the ring loader: R batch slots in ONE anonymous shared mapping that the loader's worker
    processes inherit over fork and that is page-locked for the GPU's DMA engines (hipHostRegister).  A worker collates
    its tiles straight into slot k % R -- the only host-side copy of a pixel -- and sends back a few bytes; the consumer
    issues the asynchronous H2D copy from the slot itself.  The stock DataLoader path moves every batch three times
    (collate into shared memory, pin-memory thread, H2D) and its single pin thread tops out near 8 GB/s: a
    16 k patches/s encoder eats 17 GB/s of 512 x 512 x 3 uint8 tiles + labels.
    Slot reuse: with at most `in_flight` batches dispatched ahead of the consumer, slot k % R is rewritten only after
    the consumer has fetched batch k + R - in_flight >= k + 2; the consumer waits for batch k's copy event before it
    fetches batch k + 2 (run_eval), so R = in_flight + 2 slots suffice."""

    def __init__(self, n_slots, n_max, img_shape, img_dtype, lab_shape, lab_dtype, pin=True, device=None):
        import mmap
        self.n_slots, self.n_max = int(n_slots), int(n_max)
        self.img_shape, self.img_dtype = tuple(img_shape), img_dtype
        self.lab_shape, self.lab_dtype = tuple(lab_shape), lab_dtype
        self.img_bytes = int(np.prod(img_shape)) * torch.empty((), dtype=img_dtype).element_size()
        self.lab_bytes = int(np.prod(lab_shape)) * torch.empty((), dtype=lab_dtype).element_size()
        self.lab_off = -(-self.n_max * self.img_bytes // 4096) * 4096
        self.slot_bytes = self.lab_off + -(-self.n_max * self.lab_bytes // 4096) * 4096
        self.nbytes = self.slot_bytes * self.n_slots
        self._mm = mmap.mmap(-1, self.nbytes)                     # MAP_SHARED | MAP_ANONYMOUS: shared with forked workers
        self._buf = torch.frombuffer(self._mm, dtype=torch.uint8)
        self.pinned = False                                        # True once every slot is page-locked
        self._pin_thread = None
        self._slot_ready = None                                    # one threading.Event per slot while / after pinning
        self._slots_done = 0
        self._want_pin = bool(pin) and torch.cuda.is_available()
        self._device = None
        if self._want_pin:                                          # an index for the helper thread (torch.cuda.set_device refuses a bare "cuda")
            d = torch.device(device) if device is not None else torch.device("cuda")
            self._device = d.index if d.index is not None else torch.cuda.current_device()
        self.pin_error = None

    def pin_async(self):
        """Page-lock the mapping slot by slot on a helper thread (a few GB take 0.4 - 1.2 s in all, ~20 ms per 79 MB slot): called
        AFTER the loader has forked its workers -- a fork while another thread sits inside the driver is best avoided -- it runs
        beside the production of the first batches (the workers only write into the mapping), slot 0 first, so that batch k finds
        slot k registered when it arrives.  wait_pinned(slot) before a DMA from that slot."""
        if not self._want_pin or self._pin_thread is not None or self.pinned:
            return
        import threading
        self._slot_ready = [threading.Event() for _ in range(self.n_slots)]

        def _register():
            try:
                torch.cuda.set_device(self._device)
                rt = torch.cuda.cudart()
                for s_ in range(self.n_slots):
                    rc = int(rt.cudaHostRegister(self._buf.data_ptr() + s_ * self.slot_bytes, self.slot_bytes, 0))
                    if rc != 0:
                        self.pin_error = f"hipHostRegister returned {rc} at slot {s_}"
                        break
                    self._slots_done = s_ + 1
                    self._slot_ready[s_].set()
                self.pinned = self._slots_done == self.n_slots
            except Exception as e:                                 # kept for wait_pinned(): a ring that is not page-locked still works, slowly
                self.pin_error = f"{type(e).__name__}: {e}"
            finally:
                for ev in self._slot_ready:                        # never leave a waiter behind
                    ev.set()

        self._pin_thread = threading.Thread(target=_register, name="vqae-ring-pin")
        self._pin_thread.start()

    def wait_pinned(self, slot=None):
        """Block until `slot` (default: every slot) is page-locked; warns once if the ring could not be registered."""
        if self._pin_thread is None and not self.pinned:
            self.pin_async()
        if self._slot_ready is not None and slot is not None and not self.pinned:
            self._slot_ready[slot].wait()
            if self.pin_error is None:
                return True
        if self._pin_thread is not None and (slot is None or self.pin_error is not None):
            self._pin_thread.join()
            self._pin_thread = None
            if self._want_pin and not self.pinned:
                import warnings
                warnings.warn(f"PinnedRing: the loader ring could not be page-locked ({self.pin_error}); host-to-device copies will be slower")
        return self.pinned

    def views(self, slot, n):
        """(imgs [n, *img_shape], labels [n, *lab_shape]) views of slot `slot`."""
        base = slot * self.slot_bytes
        a = self._buf[base: base + n * self.img_bytes].view(self.img_dtype).reshape((n,) + self.img_shape)
        b = self._buf[base + self.lab_off: base + self.lab_off + n * self.lab_bytes].view(self.lab_dtype).reshape((n,) + self.lab_shape)
        return a, b

    def close(self):
        if self._pin_thread is not None:
            self._pin_thread.join()
            self._pin_thread = None
        if self._slots_done:
            rt = torch.cuda.cudart()
            for s_ in range(self._slots_done):
                rt.cudaHostUnregister(self._buf.data_ptr() + s_ * self.slot_bytes)
            self._slots_done = 0
            self.pinned = False
        self._buf = None
        try:
            self._mm.close()
        except BufferError:                                        # a view is still alive somewhere: the mapping goes with it
            pass


class _RingItems(Dataset):
    """What the ring loader's workers run: the tiles of one (tagged) batch share -> the ring slot of that batch."""

    def __init__(self, dataset, ring, send_meta):
        self.dataset, self.ring, self.send_meta = dataset, ring, send_meta

    def __len__(self):
        return len(self.dataset)

    def __getitems__(self, keys):
        k, idxs = -int(keys[0]) - 1, keys[1:]
        imgs, labs = self.ring.views(k % self.ring.n_slots, len(idxs))
        metas = []
        for j, i in enumerate(idxs):
            img, lab, meta = self.dataset[i]
            imgs[j].copy_(torch.as_tensor(img))
            labs[j].copy_(torch.as_tensor(lab))
            if self.send_meta:
                metas.append(meta)
        return [(k, len(idxs), default_collate(metas) if metas else None)]


def _pool_labels(lab, out_hw):
    """adaptive_max_pool2d(labels, encoding grid) (extract_embeddings.py:127-130) on the device."""
    return ops.label_maxpool(lab.reshape(lab.shape[0], lab.shape[-2], lab.shape[-1]).to(torch.uint8), out_hw)


_COMPACT = {1: torch.uint8, 2: getattr(torch, "uint16", torch.int16), 4: torch.int32}


class _Rechunk:
    """Re-batches a stream of tile batches for the encoder: tiles go in as the loader delivers them (batches of any size), the
    encoder is called on exactly `eb` tiles at a time (the last call takes the remainder), code tiles come out in the same order and
    are handed back in the loader's batch sizes.  Why: the trunk kernels own 128 code-grid pixels = 1/8 of a tile each and a
    launch fills 512 workgroup slots, so a batch that is not a multiple of 64 tiles leaves the last round of every launch partly
    empty -- the reference's default of 100 tiles runs the encoder 6.5 % slower than 128.  The encoder is
    batch-invariant bit for bit (tests/test_configs_gpu.py), so the codes do not depend on where the stream is cut."""

    def __init__(self, enc, eb):
        self.enc, self.eb = enc, int(eb)
        self.inq, self.n_in, self.outq, self.n_out = [], 0, [], 0

    @staticmethod
    def _take(q, n):
        parts, need = [], n
        while need > 0:
            t = q[0]
            if len(t) <= need:
                parts.append(q.pop(0))
                need -= len(t)
            else:
                parts.append(t[:need])
                q[0] = t[need:]
                need = 0
        return parts[0] if len(parts) == 1 else torch.cat(parts, 0)

    def _encode(self, n):
        self.outq.append(self.enc(self._take(self.inq, n)))
        self.n_in -= n
        self.n_out += n

    def push(self, x):
        self.inq.append(x)
        self.n_in += len(x)
        while self.n_in >= self.eb:
            self._encode(self.eb)

    def flush(self):
        if self.n_in:
            self._encode(self.n_in)

    def pop(self, n):
        if self.n_out < n:
            return None
        self.n_out -= n
        return self._take(self.outq, n)


@torch.no_grad()
def run_eval(model, dataset, batch_size=100, *, autocast_dtype=_REF_AUTOCAST, num_workers=6, prefetch_factor=5,
             device=None, shard=True, encode_fn=None, pool_fn=None, loader="auto", timer=None, compact=False,
             gather_to=None, encode_batch="auto"):
    """Batched encoder pass: drop-in for run_eval (extract_embeddings.py:92-138).  Yields, per batch, the reference's pair
        ((encoding_indices, names, img_index, patch_index), (labels_pooled, names, img_index, patch_index))
    with tensors on `device`.  Defaults are the reference's: batch 100, 6 loader workers, prefetch 5, page-locked host
    buffers (:95-101), the encoder under fp16 autocast (:124-125; pass autocast_dtype=None for fp32 convolutions, or
    torch.bfloat16).

    loader: "ring" -- workers collate into a shared page-locked ring and the H2D copy starts from there (PinnedRing;
    needs num_workers > 0 and fixed-shape tensor items); "torch" -- the stock DataLoader with pin_memory, as the
    reference builds it; "auto" -- "ring" when it applies.
    compact=True returns the code tiles in the smallest unsigned dtype that holds the codebook (uint8 / uint16; what
    get_encodings stitches and downloads) instead of the reference's int64.
    encode_batch: tiles per encoder call on one GPU.  "auto" -- the next multiple of 64 when batch_size is not one (the stream of loader
    batches is re-cut for the encoder and the codes are handed back in the loader's batches, see _Rechunk; the yields are the
    same, a few batches later), else batch_size; an int forces a size; None keeps one encoder call per loader batch.

    Under torch.distributed (one process per GPU) every global batch is sharded contiguously over the ranks: a
    rank's loader reads and its GPU encodes only its share (ShardBatchSampler); the shares are re-assembled with ONE
    fixed-size all-gather per batch (dist.all_gather_shares: uint8 / uint16 code tiles + pooled label tiles in one
    buffer, no size exchange, no host sync), issued asynchronously so it overlaps the next batch's encoder pass.
    Every rank yields the full batch in the original order -- except that with gather_to=r the ranks other than r
    yield None for the two data tensors (they take part in the collective but never unpack it: get_encodings' default).

    encode_fn(imgs_on_device) -> indices [b, h, w] and pool_fn(labels_on_device, out_hw) -> [b, out_hw, out_hw]
    replace the HIP encoder / max-pool (used by the CPU tests of the sharded path; the product default has no
    CPU fallback and raises without a GPU)."""
    device = torch.device(device) if device is not None else torch.device("cuda")
    rank, ws = vdist.world() if shard else (0, 1)
    on_gpu = device.type == "cuda"
    n_codes = _num_codes(model) if model is not None else None
    code_bytes = 1 if (n_codes or 1 << 30) <= 256 else (2 if (n_codes or 1 << 30) <= 65536 else 4)
    code_dtype = _COMPACT[code_bytes]                              # what compact=True hands out (unsigned)
    wire_dtype = {1: torch.uint8, 2: torch.int16, 4: torch.int32}[code_bytes]   # same bits; torch's cat / to() know these
    native_compact = compact and encode_fn is None and hasattr(model, "with_dtype")

    def default_enc(x):
        if native_compact:                                         # NativeVQAE: codes leave the VQ kernel already compact
            nat = model.with_dtype(autocast_dtype)
            if x.dtype == torch.uint8:
                return nat.encode_u8(x, idx_dtype=code_dtype)[1]
            return nat.encode(x, "NCHW", idx_dtype=code_dtype, want_q=False, want_loss=False)[1]
        return _encode(model, x, autocast_dtype)

    enc = encode_fn or default_enc
    pool = pool_fn or _pool_labels
    geometry = _has_geometry(dataset)
    n_items = len(dataset)
    n_batches = -(-n_items // batch_size)
    factor = _factor(model) if encode_fn is None else None
    cap = vdist.gather_capacity(batch_size, ws)
    want_data = gather_to is None or gather_to == rank

    # ---- the loader ------------------------------------------------------------------------------------------------------
    use_ring = loader == "ring" or (loader == "auto" and num_workers > 0 and n_items > 0)
    ring, probe = None, None
    if use_ring:
        probe = dataset[0]
        ok = isinstance(probe[0], torch.Tensor) and isinstance(probe[1], torch.Tensor)
        if not ok and loader == "ring":
            raise TypeError("run_eval(loader='ring'): dataset items must be (tensor, tensor, meta)")
        use_ring = ok
    sampler = ShardBatchSampler(n_items, batch_size, rank, ws, tag_batches=use_ring)
    extra = {"prefetch_factor": prefetch_factor} if num_workers else {}
    if use_ring:
        in_flight = max(1, num_workers) * (prefetch_factor if num_workers else 1)
        with _stage(timer, "host", "ring_setup"):
            ring = PinnedRing(in_flight + 2, cap, probe[0].shape, probe[0].dtype, probe[1].shape, probe[1].dtype, pin=on_gpu,
                              device=device if on_gpu else None)
        if num_workers:
            extra["multiprocessing_context"] = "fork"              # the workers must inherit the ring's mapping
        dl = DataLoader(_RingItems(dataset, ring, send_meta=not geometry), batch_sampler=sampler,
                        collate_fn=lambda b: b[0], pin_memory=False, num_workers=num_workers, **extra)
    else:
        dl = DataLoader(dataset, batch_sampler=sampler, collate_fn=_collate, pin_memory=on_gpu, num_workers=num_workers,
                        **extra)
    labels_dtype = getattr(dataset, "labels_dtype", None) or (probe[1].dtype if probe is not None else None)

    def meta_of(k, n, collated):
        """(img_index, patch_index, image paths, mask paths) of global batch k on every rank"""
        b0 = k * batch_size
        if geometry:
            return batch_meta(dataset, b0, b0 + n)
        if ws == 1:
            return collated
        # no geometry attributes (a Subset, a wrapper ...): the metadata travels with the tiles -- object gather (host sync)
        import torch.distributed as tdist
        parts = [None] * ws
        tdist.all_gather_object(parts, collated)
        parts = [p for p in parts if p is not None]
        return (torch.cat([torch.as_tensor(p[0]) for p in parts]), torch.cat([torch.as_tensor(p[1]) for p in parts]),
                sum((list(p[2]) for p in parts), []), sum((list(p[3]) for p in parts), []))

    def emit(idx, pooled, meta):
        img_index, patch_index, img_path, label_path = meta
        return ((data, list(map(_extract_path, paths)), img_index, patch_index)
                for data, paths in ((idx, img_path), (pooled, label_path)))

    def finish(item):
        """gathered buffer -> the reference's generator of two tuples, for one batch"""
        (out, work, n, th, tw, meta, ldt) = item
        if work is not None:
            work.wait()
        if not want_data:
            return emit(None, None, meta)
        nb = th * tw * code_bytes
        idx_parts, pool_parts = [], []
        for r in range(ws):
            lo, hi = vdist.shard_range(n, r, ws)
            rows = out[r, : hi - lo]
            idx_parts.append(rows[:, :nb].contiguous().view(wire_dtype).reshape(hi - lo, th, tw))
            pool_parts.append(rows[:, nb:].reshape(hi - lo, th, tw))
        idx, pooled = torch.cat(idx_parts, 0), torch.cat(pool_parts, 0)
        if compact:
            idx = idx.view(code_dtype)
        else:
            idx = idx.to(torch.int64)
            if code_bytes == 2:
                idx = idx & 0xFFFF                                 # uint16 codes travelled as int16 bit patterns
        return emit(idx, pooled.to(ldt), meta)

    pending, grid_hw, copied, encoded, depth = None, None, [], [], 3
    eb = encode_batch
    if eb == "auto":
        # the next multiple of 64 tiles: whole rounds of workgroups, hardly more latency (measured at batch 100, cfg A f16, 100 k tiles:
        # encoder 6.97 s -> 6.65 s with 128 or 256 per call; end to end 7.91 s -> 7.62 s with 128, 7.95 - 8.04 s with 256: burstier)
        per_call = batch_size if ws == 1 else cap                  # tiles this rank encodes per loader batch (cap: its largest share)
        eb = -(-per_call // 64) * 64 if (on_gpu and encode_fn is None and per_call % 64 != 0 and per_call < 512) else None
    rq, waiting = None, []                                         # re-cut stream; batches whose codes are still due

    def _launch_share(item):
        """sharded: my share of one batch (codes + pooled labels, `cap` rows) -> asynchronous all-gather; None while the re-cut
        encoder stream has not produced the batch's codes yet"""
        idx_s, n_s, pooled_s, (n_b, meta_b) = item
        th_, tw_ = grid_hw
        if idx_s is None:
            idx_s = rq.pop(n_s)
            if idx_s is None:
                return None
        if idx_s is False:                                         # empty share
            mine = torch.zeros((cap, th_ * tw_ * (code_bytes + 1)), dtype=torch.uint8, device=device)
        else:
            with _stage(timer, "gpu", "pack"):
                nb = th_ * tw_
                codes = (idx_s if idx_s.element_size() == code_bytes else idx_s.to(wire_dtype))
                codes = codes.reshape(idx_s.shape[0], nb).contiguous().view(torch.uint8)
                mine = torch.zeros((cap, codes.shape[1] + nb), dtype=torch.uint8, device=device)
                mine[: idx_s.shape[0], : codes.shape[1]] = codes
                mine[: idx_s.shape[0], codes.shape[1]:] = pooled_s.reshape(idx_s.shape[0], nb).to(torch.uint8)
        with _stage(timer, "host", "gather_launch"):
            out, work = vdist.all_gather_shares(mine, async_op=True)
        return (out, work, n_b, th_, tw_, meta_b, labels_dtype)
    if on_gpu:
        # the copy stream gets a high-priority hardware queue of its own: an ordinary stream may land on the queue the encoder's
        # launches sit in (HIP deals its few hardware queues round-robin) and every copy then waits behind up to `depth` batches of kernels
        main_stream, copy_stream = torch.cuda.current_stream(device), torch.cuda.Stream(device, priority=-1)
    with _stage(timer, "host", "loader_start"):
        it = iter(dl)
    if ring is not None:
        ring.pin_async()
    try:
        for k in range(n_batches):
            with _stage(timer, "host", "loader_wait"):
                batch = next(it)
            n = min(batch_size, n_items - k * batch_size)
            imgs = labels = collated = None
            if use_ring:
                _, n_mine, collated = batch
                if n_mine:
                    imgs, labels = ring.views(k % ring.n_slots, n_mine)
            elif batch is not None:
                imgs, labels, collated = batch
            if imgs is not None:
                if on_gpu:
                    # H2D on its own stream: batch k + 1 goes up while batch k is being encoded (PCIe moves 105 MB per batch of
                    # 100 uint8 512 x 512 tiles + labels: 1.9 ms of the 7 ms the encoder needs).  The host may run at most
                    # `depth` batches ahead of the encoder (device memory for the inputs stays bounded).
                    if len(encoded) >= depth:
                        with _stage(timer, "host", "encoder_backpressure"):
                            encoded.pop(0).synchronize()
                    if use_ring and k < ring.n_slots:              # the slots are page-locked in order, beside the first batches
                        with _stage(timer, "host", "ring_pin_wait"):
                            ring.wait_pinned(k % ring.n_slots)
                    with torch.cuda.stream(copy_stream):
                        with _stage(timer, "gpu", "h2d"):
                            x = imgs.to(device, non_blocking=True)
                            lab = labels.to(device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    main_stream.wait_event(ev)
                    x.record_stream(main_stream)
                    lab.record_stream(main_stream)
                    if use_ring:                                   # slot k is free again once this copy has run
                        copied.append(ev)
                        if len(copied) > 1:
                            with _stage(timer, "host", "ring_backpressure"):
                                copied.pop(0).synchronize()
                else:
                    x, lab = imgs.to(device), labels.to(device)
                if eb and grid_hw is not None:                     # (the first batch is encoded on its own: it tells the grid size)
                    if rq is None:
                        rq = _Rechunk(enc, eb)
                    with _stage(timer, "gpu", "encode"):
                        rq.push(x)
                    idx = None
                    th, tw = grid_hw
                else:
                    with _stage(timer, "gpu", "encode"):
                        idx = enc(x)
                    th, tw = int(idx.shape[-2]), int(idx.shape[-1])
                if labels_dtype is None:
                    labels_dtype = labels.dtype
                with _stage(timer, "gpu", "label_pool"):
                    pooled = pool(lab, tw)
                if on_gpu:
                    ev = torch.cuda.Event()
                    ev.record(main_stream)
                    encoded.append(ev)
                grid_hw = (th, tw)
            if ws == 1:                                            # the reference's single-GPU loop (:117-138)
                waiting.append((idx, n, pooled, meta_of(k, n, collated)))
                while waiting:
                    idx_w, n_w, pooled_w, meta_w = waiting[0]
                    if idx_w is None:
                        idx_w = rq.pop(n_w)
                        if idx_w is None:
                            break
                    waiting.pop(0)
                    if compact and idx_w.dtype == torch.int64:
                        idx_w = idx_w.to(wire_dtype).view(code_dtype)
                    yield emit(idx_w, pooled_w.reshape(idx_w.shape).to(labels_dtype), meta_w)
                continue
            # ---- sharded: pack my share, launch the gather, and only then hand out the PREVIOUS batch ---------------------
            # (with a re-cut encoder stream a batch is launched once its codes exist: every rank launches the gathers in batch
            # order, one or a few iterations later than it read the batch -- the collectives still pair up by order)
            meta = meta_of(k, n, collated)
            if imgs is None:                                       # empty share (a last batch shorter than the world size)
                if grid_hw is None:                                # ... before this rank ever encoded a tile
                    ps = getattr(dataset, "patch_size", None)
                    assert factor is not None and ps is not None, "run_eval: an empty share needs dataset.patch_size and a model factor"
                    grid_hw = (int(ps[0]) // factor, int(ps[1]) // factor)
                waiting.append((False, 0, None, (n, meta)))
            else:
                waiting.append((idx, int(x.shape[0]), pooled, (n, meta)))
            if labels_dtype is None:                               # no tile of mine yet: what every other rank sees in its batches
                labels_dtype = dataset[0][1].dtype
            while waiting:
                launched = _launch_share(waiting[0])
                if launched is None:
                    break
                waiting.pop(0)
                if pending is not None:
                    yield finish(pending)
                pending = launched
        if ws > 1 and rq is not None:                              # the tail of the re-cut stream, sharded
            with _stage(timer, "gpu", "encode"):
                rq.flush()
            for item in waiting:
                launched = _launch_share(item)
                if pending is not None:
                    yield finish(pending)
                pending = launched
            waiting = []
        if pending is not None:
            yield finish(pending)
        if ws == 1 and rq is not None:                             # the tail of the re-cut stream
            with _stage(timer, "gpu", "encode"):
                rq.flush()
            for idx_w, n_w, pooled_w, meta_w in waiting:
                idx_w = rq.pop(n_w) if idx_w is None else idx_w
                if compact and idx_w.dtype == torch.int64:
                    idx_w = idx_w.to(wire_dtype).view(code_dtype)
                yield emit(idx_w, pooled_w.reshape(idx_w.shape).to(labels_dtype), meta_w)
            waiting = []
    finally:
        # Joining the worker processes (each unmaps the ring) and un-registering the ring take 0.5 - 1 s together: once the device has
        # drained the copies that read the ring, both run on a helper thread beside whatever the caller does next (the stitch / HDF5
        # write of the last slide, the next dataset).  The thread is not a daemon: interpreter exit waits for it.
        if ring is not None and on_gpu:
            with _stage(timer, "host", "drain"):
                torch.cuda.synchronize()
        holder = [it]
        del it

        def _teardown(holder=holder, ring=ring):
            holder.clear()                                         # drops the last reference: DataLoader shuts its workers down
            if ring is not None:
                ring.close()

        started = False
        if ring is not None and on_gpu:
            import threading
            try:
                threading.Thread(target=_teardown, name="vqae-loader-teardown").start()
                started = True
            except RuntimeError:                                   # interpreter shutting down (a generator finalised late): do it here
                pass
        if not started:
            _teardown()


def _stitch(sel, rc, grid):
    return ops.stitch_tiles(sel, rc, grid)


def _runs(img_idx_np):
    """[(lo, hi)] maximal runs of equal slide index in a batch (tiles arrive in dataset order: one run per slide)."""
    if len(img_idx_np) == 0:
        return []
    cuts = np.flatnonzero(np.diff(img_idx_np)) + 1
    edges = np.concatenate(([0], cuts, [len(img_idx_np)]))
    return list(zip(edges[:-1].tolist(), edges[1:].tolist()))


def get_encodings(model, dataset, batch_size=100, stitch_fn=None, replicate=False, timer=None,
                  **kw) -> Iterator[Tuple[str, Optional[np.ndarray]]]:
    """Stitch code tiles into one `[32*rows, 32*cols]` grid per slide on the device and yield
    `(name, ndarray)` -- cast to the lowest dtype -- as soon as every tile of a slide has been seen
    (extract_embeddings.py:43-89).  `stitch_fn(tiles, rc, grid)` replaces the HIP scatter (CPU tests only).

    No host synchronisation per batch: the tiles of a slide are a contiguous run of the batch (a view, not a
    boolean-mask gather), their (row, col) go up in one page-locked copy per batch, the slide grid lives on the device
    in the codes' own width (uint8 for <= 256 codes: a 100 k-tile slide downloads 102 MB, not 819 MB of int64) and comes
    down once, when the slide is complete.  Under torch.distributed only rank 0 stitches and downloads (it is the only
    writer, save_encodings*); the other ranks drive the same collectives and yield `(name, None)`.  replicate=True
    restores a full copy on every rank."""
    stitch = stitch_fn or _stitch
    rank, ws = vdist.world() if kw.get("shard", True) else (0, 1)
    active = replicate or ws == 1 or rank == 0
    arrays, counts = {}, {}
    for ret_values in run_eval(model, dataset, batch_size=batch_size, compact=True, timer=timer,
                               gather_to=None if (replicate or ws == 1) else 0, **kw):
        for (encodings, names, img_idx, patch_idx) in ret_values:
            img_idx_np = np.asarray(img_idx)
            rc_dev = None
            for lo, hi in _runs(img_idx_np):
                name, image_index = names[lo], int(img_idx_np[lo])
                if name not in counts:
                    counts[name] = int(dataset._lengths[image_index])
                    if active:
                        th, tw = int(encodings.shape[1]), int(encodings.shape[2])
                        r, c = (int(v) for v in dataset._sizes[image_index])
                        arrays[name] = torch.empty((r * th, c * tw), dtype=encodings.dtype, device=encodings.device)
                if active:
                    with _stage(timer, "gpu", "stitch"):
                        if rc_dev is None:                             # one upload per batch, page-locked when there is a GPU
                            rc_host = torch.as_tensor(np.asarray(patch_idx), dtype=torch.int32)
                            if encodings.is_cuda:
                                rc_host = rc_host.pin_memory()
                            rc_dev = rc_host.to(encodings.device, non_blocking=True)
                        stitch(encodings[lo:hi], rc_dev[lo:hi], arrays[name])            # device scatter (:83-84)
                counts[name] -= hi - lo
                if counts[name] == 0:
                    counts.pop(name)
                    if not active:
                        yield str(name), None
                        continue
                    with _stage(timer, "host", "d2h"):
                        host = arrays.pop(name).cpu().numpy()
                    with _stage(timer, "host", "cast_lowest"):
                        host = cast_to_lowest_dtype(host)
                    yield str(name), host


def save_encodings(root, model, dataset, **kw):
    """`np.save(<root>/encodings/<images|masks>/<stem>.npy)` per slide (extract_embeddings.py:179-185).
    Under torch.distributed only rank 0 writes."""
    rank, _ = vdist.world()
    written = []
    for name, array in get_encodings(model, dataset, **kw):
        if rank == 0 and array is not None:
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
    timer = kw.get("timer")
    try:
        for name, array in get_encodings(model, dataset, **kw):
            if writer is not None and array is not None:
                group, _, stem = name.rpartition("/")
                with _stage(timer, "host", "hdf5_write"):
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
