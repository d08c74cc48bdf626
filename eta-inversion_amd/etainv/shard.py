"""Batch sharding of an image list over ranks (one process per GPU) and the final gather of edited latents.

The reference parallelises by launching one OS process per evaluation config pinned with CUDA_VISIBLE_DEVICES
(eval.py:112-183) and has no collective.  Every image's invert -> edit is independent, so images are
dealt round-robin to ranks; the only exchange is one all_gather of the (n,4,L,L) latents (32 KiB per image at L = 64),
RCCL over xGMI on the GPU box (backend "nccl"), gloo in the CPU tests."""
from typing import List

import torch


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """image i goes to rank i % world (700 images over 8 ranks -> 88/87 per rank)"""
    return list(range(rank, n_items, world))


def gather_latents(local: torch.Tensor, n_items: int, rank: int, world: int, group=None) -> torch.Tensor:
    """local: (n_local, ...) results for shard_indices(n_items, rank, world), in that order.  Returns all n_items results
    in image order on every rank.  Shards are padded to equal length for the collective."""
    if world == 1:
        return local
    import torch.distributed as dist
    per = (n_items + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        out[idx] = parts[r][: len(idx)]
    return out
