"""One process per GPU: patch-batch sharding + the path's only collective.

Every patch is independent in eval mode (no batch statistics, per-patch circular padding, read-only
codebook), so the data path needs no collective: rank r encodes a contiguous share of
each patch batch with a replicated handle.  The single exchange step is the all-gather that
reassembles per-slide code grids (uint8/uint16/int32 tiles, <= 512 KB per rank per batch), issued
through torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU
tests.  The reference itself extracts on one GPU only (extract_embeddings.py:106).
"""
import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Initialise from torchrun's RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (rendezvous on 127.0.0.1)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, ws, local


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous share [lo, hi) of n items for `rank`; the first n % world ranks get one extra."""
    q, r = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_capacity(batch_size: int, world_size: int) -> int:
    """Tiles a rank can own of one global batch (= the share of rank 0)."""
    return -(-batch_size // world_size)


def all_gather_shares(mine: torch.Tensor, async_op: bool = False):
    """The data path's only collective.  `mine` [cap, row_bytes] uint8: this rank's share of one batch (code tiles +
    pooled label tiles, packed), zero-padded to the fixed per-rank capacity -> (out [world, cap, row_bytes], work).
    Sizes are fixed by (batch_size, world_size) and every rank knows every share's length from shard_range, so there
    is no size exchange and no host sync; with async_op the gather (RCCL: its own stream) overlaps the next batch."""
    rank, ws = world()
    out = mine.new_empty((ws,) + tuple(mine.shape))
    if ws == 1:
        out[0].copy_(mine)
        return out, None
    if mine.is_cuda and dist.get_backend() == "gloo":
        # gloo moves host memory: rehearsals of the sharded GPU path on a box without RCCL peers (several ranks on one GPU,
        # tests/test_driver_gpu.py) stage the (small) share through the host; under "nccl" (RCCL) the tensors stay in HBM
        host = torch.empty((ws,) + tuple(mine.shape), dtype=mine.dtype)
        dist.all_gather_into_tensor(host.view(-1), mine.contiguous().view(-1).cpu())
        out.copy_(host)
        return out, None
    work = dist.all_gather_into_tensor(out.view(-1), mine.contiguous().view(-1), async_op=async_op)
    return out, work


def all_gather_ragged(x: torch.Tensor) -> List[torch.Tensor]:
    """All-gather tensors whose first dimension differs per rank (last, short batch of a slide).
    One size exchange + one padded all_gather_into_tensor; returns the per-rank tensors in rank order."""
    rank, ws = world()
    if ws == 1:
        return [x]
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = x
    if x.shape[0] < m:
        pad = torch.cat([x, x.new_zeros((m - x.shape[0],) + tuple(x.shape[1:]))], 0)
    out = x.new_empty((ws * m,) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous())
    return [out[i * m: i * m + sizes[i]] for i in range(ws)]


def all_gather_codes(idx: torch.Tensor, meta: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """idx [b, h, w] code tiles and meta [b, 3] = (slide, row, col) of this rank's share ->
    concatenation over ranks in rank order (= original patch order for contiguous shards)."""
    parts = all_gather_ragged(idx)
    metas = all_gather_ragged(meta)
    return torch.cat(parts, 0), torch.cat(metas, 0)
