"""Multi-GPU execution of the sampling path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm).

Sampling chains are independent per volume (GroupNorm and attention are per-sample: reference diffusion_model_unet.py:623,
407-415), so the path shards by *unit = one volume*: every rank samples its own contiguous slice of the unit list and there is
NO collective on the data path. The only communication is the optional gather of finished volumes to one rank. (The reference
ships no multi-GPU sampling code at all; its only distributed example is the DDP training tutorial.)"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch


def shard_range(n_units: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of `n_units` for `rank` (the first n_units % world ranks get one extra)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_units, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def unit_seed(base_seed: int, unit: int) -> int:
    """Per-volume RNG seed: a volume's noise depends on its global index only, never on the rank that samples it."""
    return (int(base_seed) * 1_000_003 + int(unit)) % (2**63 - 1)


def unit_noise(shape: Sequence[int], base_seed: int, unit: int, dtype=torch.float32) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=torch.Generator().manual_seed(unit_seed(base_seed, unit))).to(dtype)


def sample_units(sample_one: Callable[[torch.Tensor], torch.Tensor], n_units: int, noise_shape: Sequence[int], base_seed: int,
                 rank: int, world_size: int, device=None, dtype=torch.float32) -> Tuple[List[int], List[torch.Tensor]]:
    """Run `sample_one(noise)` (e.g. `lambda z: inferer.sample(z, model, scheduler, verbose=False)`) on this rank's units."""
    begin, end = shard_range(n_units, rank, world_size)
    ids, outs = [], []
    for u in range(begin, end):
        z = unit_noise(noise_shape, base_seed, u, dtype)
        if device is not None:
            z = z.to(device)
        ids.append(u)
        outs.append(sample_one(z))
    return ids, outs


def gather_units(ids: List[int], outs: List[torch.Tensor], n_units: int, group=None, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Collect every rank's finished volumes on rank `dst`, ordered by global unit index (None on the other ranks).
    Works with any backend (tensors are exchanged as objects on the host: this is result collection, not the data path)."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return list(outs)
    world = dist.get_world_size(group)
    payload = [(i, o.detach().cpu()) for i, o in zip(ids, outs)]
    gathered: Optional[List] = [None] * world if dist.get_rank(group) == dst else None
    dist.gather_object(payload, gathered, dst=dst, group=group)
    if dist.get_rank(group) != dst:
        return None
    table = {}
    for part in gathered:
        for i, o in part:
            if i in table:
                raise RuntimeError(f"unit {i} was produced twice")
            table[i] = o
    if sorted(table) != list(range(n_units)):
        raise RuntimeError("some units are missing after the gather")
    return [table[i] for i in range(n_units)]
