"""Data-parallel serving of the denoise loop: one process per GPU, requests sharded across ranks, finished
latents gathered over RCCL/xGMI.

The reference creates a `_DP` group but never uses it — every rank runs the same request
(vllm_omni/diffusion/distributed/parallel_state.py:661-668, scheduler.py:55-62).  Here DP is the
scaling axis: weights (41 GB bf16) replicate into each 288 GB GPU, a request's denoise loop
touches only its own latents, so there is NO collective inside the loop.  The only exchange is the end-of-batch
gather of finished packed latents ([n, S_img, 64] bf16 = 512 KB per 1024^2 image): one `all_gather_into_tensor`
(RCCL on GPU — backend "nccl" IS RCCL on ROCm — gloo in the CPU tests), latency-bound, one hop on the xGMI mesh.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None, timeout_s: int | None = None, local_device: int | None = None) -> tuple[int, int, int]:
    """env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT).  Returns (rank, world, local).
    `local_device` overrides LOCAL_RANK as the device index (explicit rank -> device maps)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank))) if local_device is None else int(local_device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if timeout_s:
            import datetime

            kw["timeout"] = datetime.timedelta(seconds=timeout_s)
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kw)
    return rank, world, local


def init_sp_groups(world: int, degree: int, rank: int):
    """Partition the world into world/degree sequence-parallel groups of CONSECUTIVE ranks (rank r: group r // degree,
    position r % degree — neighbours on the xGMI mesh) and return (this rank's group, dp_rank, dp_world).  Every rank must
    call this (dist.new_group is collective over the world)."""
    if degree <= 1:
        return None, rank, world
    if world % degree:
        raise ValueError(f"world size {world} is not divisible by ulysses_degree {degree}")
    mine = None
    for g in range(world // degree):
        ranks = list(range(g * degree, (g + 1) * degree))
        grp = dist.new_group(ranks=ranks) if dist.is_initialized() and world > 1 else None
        if rank in ranks:
            mine = grp
    return mine, rank // degree, world // degree


def shard_requests(costs: list[float], world: int) -> list[list[int]]:
    """Greedy least-loaded assignment of request indices to ranks (cost = denoise steps x tokens).
    Deterministic on every rank, so no coordination message is needed."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += costs[i]
    for lst in out:
        lst.sort()
    return out


def any_rank_failed(failed: bool, device=None, group=None) -> bool:
    """Agree on a failure flag before a collective: MAX-all-reduce of one int32 (world 1: the local flag)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(failed)
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def gather_latents(local: torch.Tensor, counts: list[int], group=None) -> torch.Tensor:
    """All ranks contribute `local` [counts[rank], S, C]; every rank receives [sum(counts), S, C] in rank order.
    One padded all_gather_into_tensor (ranks with fewer items pad to max(counts))."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    rank = dist.get_rank(group)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    if local.is_cuda and dist.get_backend(group) != "nccl":
        # a gloo group (the CPU tests; `bench.py --dist-backend gloo`, which exercises the N-rank control flow on ONE device)
        # has no all_gather for device tensors: stage through the host
        return gather_latents(local.cpu(), counts, group).to(local.device)
    mx = max(counts)
    S, Cc = local.shape[1], local.shape[2]
    send = local.new_zeros((mx, S, Cc))
    send[: counts[rank]] = local
    recv = local.new_empty((world * mx, S, Cc))
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = [recv[r * mx: r * mx + counts[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def unshard(gathered: torch.Tensor, assignment: list[list[int]]) -> list[torch.Tensor]:
    """Undo shard_requests: gathered rows are in rank order; return them in request order."""
    n = sum(len(a) for a in assignment)
    out: list[torch.Tensor | None] = [None] * n
    k = 0
    for lst in assignment:
        for i in lst:
            out[i] = gathered[k]
            k += 1
    return out  # type: ignore[return-value]


def unshard_indexed(gathered: torch.Tensor, assignment: list[list[int]]) -> dict[int, torch.Tensor]:
    """As `unshard` for an assignment over a SUBSET of the request indices (one resolution of a mixed batch): {index: row}."""
    out: dict[int, torch.Tensor] = {}
    k = 0
    for lst in assignment:
        for i in lst:
            out[i] = gathered[k]
            k += 1
    return out
