"""Multi-GPU batch split: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  Utterances are independent, so the decode
path itself needs NO collective: each rank decodes a contiguous slice of the batch with its own
replica of the weights and its own KV cache.  The only exchange is the final gather of the result
codes (B/n x G x 8 ids per rank, < 4 MB): one all_gather, latency-bound, off the critical path.
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str | None = None):
    """Initialises the default group from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split; the first (n_items % world) ranks take one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_codes(local: Sequence[torch.Tensor], n_total: int, Q: int = 8, device=None) -> List[torch.Tensor] | None:
    """All ranks contribute their utterances' (G_b, Q) code matrices; every rank gets the full list in
    global order.  Codes travel as int16 (ids < 1025), padded to the longest utterance; the lengths travel
    separately as int32 (an utterance may be longer than an int16 can count: cap 16 * S + 1)."""
    rank, _, world = env_rank_world()
    if world == 1 or not dist.is_initialized():
        return list(local)
    device = device if device is not None else (local[0].device if len(local) else torch.device("cpu"))
    spans = [shard_range(n_total, r, world) for r in range(world)]
    per_rank = max(hi - lo for lo, hi in spans)
    assert len(local) == spans[rank][1] - spans[rank][0], "rank decoded a different number of utterances than its shard"
    lens = torch.zeros(per_rank, dtype=torch.int32, device=device)
    for i, t in enumerate(local):
        assert t.dim() == 2 and t.shape[1] == Q, "each utterance is a (G, Q) code matrix"
        lens[i] = t.shape[0]
    all_lens = torch.empty(world * per_rank, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(all_lens, lens)
    all_lens = all_lens.view(world, per_rank).cpu()
    gmax = max(int(all_lens.max().item()), 1)
    buf = torch.zeros((per_rank, gmax, Q), dtype=torch.int16, device=device)
    for i, t in enumerate(local):
        buf[i, : t.shape[0]] = t.to(device=device, dtype=torch.int16)
    out = torch.empty((world * per_rank, gmax, Q), dtype=torch.int16, device=device)
    # neither NCCL/RCCL nor gloo has an int16 collective type: ship the same bytes as uint8
    dist.all_gather_into_tensor(out.view(torch.uint8), buf.view(torch.uint8))
    out = out.view(world, per_rank, gmax, Q)
    res: List[torch.Tensor] = []
    for r, (lo, hi) in enumerate(spans):
        for i in range(hi - lo):
            res.append(out[r, i, : int(all_lens[r, i])].to(torch.int64))
    return res


def decode_sharded(decode_fn: Callable[[int, int], List[torch.Tensor]], n_total: int, Q: int = 8, device=None):
    """decode_fn(lo, hi) decodes global utterances [lo, hi) on this rank's GPU; returns the gathered list."""
    rank, _, world = env_rank_world()
    lo, hi = shard_range(n_total, rank, world)
    local = decode_fn(lo, hi)
    return gather_codes(local, n_total, Q, device)
