"""Batch-sharded multi-GPU inference: one process per GPU, no data-path collective.

The reference shards its sample list with accelerate's `PartialState().split_between_processes`
(examples/brushnet/test_brushnet.py:163-168): a static contiguous split where the first `n % world` ranks get
one extra item and nothing is communicated.  `shard_range` reproduces that partition; the only
collectives used by the harness are the barrier and the max-over-ranks of the elapsed time that bench.py's
contract asks for (RCCL when the tensors are on the GPU, gloo on the CPU for tests).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple, TypeVar

import torch
import torch.distributed as dist

T = TypeVar("T")


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(backend: str = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of this rank's contiguous share (accelerate.PartialState.split_between_processes)."""
    per, extra = divmod(n_items, world)
    start = rank * per + min(rank, extra)
    return start, start + per + (1 if rank < extra else 0)


def shard(items: Sequence[T], rank: int, world: int) -> List[T]:
    a, b = shard_range(len(items), rank, world)
    return list(items[a:b])


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
