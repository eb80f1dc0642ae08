"""Multi-GPU layout of the path: one process per GPU, the batch (independent polynomials / ciphertexts / PIR shards)
is split across ranks and nothing on the data path crosses GPUs -- every PolyRq operation of the reference is
per-polynomial (the reference itself is single-process, Sources/HomomorphicEncryption has no
communication layer).  The only collective is the optional gather of finished shards (RCCL all-gather on GPUs, gloo in
the CPU tests) and the max-over-ranks reduction of the benchmark clock.

No compute happens here: these helpers only decide who owns which polynomials and move finished results.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def rank_and_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) outside a launcher."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced partition of `total` polynomials: rank r owns [begin, end).  The first total % world ranks
    own one more; an empty shard (total < world) is legal and must be a no-op for every kernel."""
    if world <= 0 or not 0 <= rank < world or total < 0:
        raise ValueError(f"bad shard request: total={total} world={world} rank={rank}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int):
    return [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]


def gather_shards(local, total: int, group=None):
    """All-gather the per-rank result shards (dim 0 = polynomials) into the full batch on every rank.
    Ragged shards are padded to the largest one for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.shape[0] != total:
            raise ValueError("single-rank gather: the local shard must be the whole batch")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} polynomials, its shard is {sizes[rank]}")
    largest = max(sizes)
    padded = local
    if local.shape[0] != largest:
        padded = torch.zeros((largest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    out = torch.empty((world, largest) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), padded.contiguous().view(-1), group=group)
    if all(s == largest for s in sizes):
        return out.view((total,) + tuple(local.shape[1:]))
    return torch.cat([out[r, : sizes[r]] for r in range(world)], dim=0)


def max_over_ranks(seconds: float, device: Optional[object] = None, group=None) -> float:
    """The job is as slow as its slowest rank: reduce a locally measured duration with MAX."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
