"""Batch sharding over ranks: clips are independent, so the STLT forward shards over the batch with no data-path
collective.  One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm, "gloo" is used
for the CPU tests.  Collectives appear only where results are merged (logits gather, metric counters)."""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). -> (rank, world)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of n clips: the first n % world ranks get one extra clip."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """This rank's contiguous slice of every per-clip tensor of a collated batch (non-tensor entries pass through)."""
    n = batch["categories"].shape[0]
    lo, hi = shard_bounds(n, rank, world)
    out = {}
    for k, v in batch.items():
        if world > 1 and k in ("num_real_tokens", "num_real_frames"):
            continue  # row counts of the GLOBAL batch (collate.real_counts): a shard reads its own back instead
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n:
            out[k] = v[lo:hi]
        elif isinstance(v, (list, tuple)) and len(v) == n:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def gather_rows(local: torch.Tensor, n_total: int, world: int) -> torch.Tensor:
    """all_gather of per-clip rows from uneven contiguous shards back into batch order. -> (n_total, ...)"""
    if world == 1:
        return local
    per = (n_total + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts, dim=0)


def all_reduce_sum_(t: torch.Tensor, world: int) -> torch.Tensor:
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
