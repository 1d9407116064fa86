"""Sequence <-> head resharding collectives for Ulysses sequence parallelism — the functions of
vllm_omni/diffusion/distributed/comm.py:12-221 (`all_to_all_4D`, `all_to_all_5D`, `SeqAllToAll4D/5D`).

One primitive does all of them: `swap_shard_axes(x, gather, scatter, group)` — every rank holds the full `scatter` axis and
1/P of the `gather` axis; afterwards it holds the full `gather` axis and its 1/P slice of the `scatter` axis.  The tensor is
cut into P blocks along `scatter`, the blocks become the leading axis, ONE `all_to_all_single` moves block j to rank j, and the
received blocks (one per source rank = one per `gather` shard, in rank order) are laid along `gather`.  On a fully connected
xGMI node that is a single step of P-1 peer writes.  Runs on RCCL (GPU) and gloo (CPU tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def _world(group) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def swap_shard_axes(x: torch.Tensor, gather: int, scatter: int, group=None, use_sync: bool = False) -> torch.Tensor:
    P = _world(group)
    if P == 1:
        return x
    if x.shape[scatter] % P:
        raise ValueError(f"axis {scatter} of size {x.shape[scatter]} is not divisible by the group size {P}")
    shp = list(x.shape)
    blk = shp[scatter] // P
    # [.., scatter=P*blk, ..] -> [P, .., blk, ..]
    send = x.reshape(shp[:scatter] + [P, blk] + shp[scatter + 1:]).movedim(scatter, 0).contiguous()
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    if use_sync and x.is_cuda:
        torch.cuda.synchronize()
    # recv[j] = rank j's shard of `gather`, restricted to my block of `scatter`: lay the P pieces along `gather`
    out = recv.movedim(0, gather)                              # [.., P, gather_loc, ..]: P lands right before gather's slot
    shp2 = list(out.shape)
    return out.reshape(shp2[:gather] + [shp2[gather] * shp2[gather + 1]] + shp2[gather + 2:]).contiguous()


def all_to_all_4D(input: torch.Tensor, scatter_idx: int = 2, gather_idx: int = 1, group=None, use_sync: bool = False):
    """(bs, seq/P, heads, hd) -> (bs, seq, heads/P, hd) for (scatter 2, gather 1); the reverse for (scatter 1, gather 2)."""
    if input.dim() != 4:
        raise ValueError(f"input must be 4D tensor, got {input.dim()} and shape {tuple(input.shape)}")
    if (scatter_idx, gather_idx) not in ((2, 1), (1, 2)):
        raise RuntimeError("scatter_idx must be 1 or 2 and gather_idx must be 1 or 2")
    return swap_shard_axes(input, gather=gather_idx, scatter=scatter_idx, group=group, use_sync=use_sync)


def all_to_all_5D(input: torch.Tensor, scatter_idx: int = 3, gather_idx: int = 1, group=None, use_sync: bool = False):
    """Fused q/k/v: (bs, seq/P, 3, heads, hd) -> (bs, seq, 3, heads/P, hd) for (scatter 3, gather 1); reverse for (1, 3)."""
    if input.dim() != 5:
        raise ValueError(f"input must be 5D tensor, got {input.dim()} and shape {tuple(input.shape)}")
    if (scatter_idx, gather_idx) not in ((3, 1), (1, 3)):
        raise RuntimeError("scatter_idx must be 1 or 3 and gather_idx must be 1 or 3")
    return swap_shard_axes(input, gather=gather_idx, scatter=scatter_idx, group=group, use_sync=use_sync)


class SeqAllToAll4D:
    """Call-compatible with the reference's autograd.Function (`SeqAllToAll4D.apply(group, x, scatter, gather, use_sync)`);
    inference only, so no backward."""

    @staticmethod
    def apply(group, input, scatter_idx: int, gather_idx: int, use_sync: bool = False):
        return all_to_all_4D(input, scatter_idx, gather_idx, group=group, use_sync=use_sync)

    @staticmethod
    def forward(ctx, group, input, scatter_idx: int, gather_idx: int, use_sync: bool = False):
        return all_to_all_4D(input, scatter_idx, gather_idx, group=group, use_sync=use_sync)


class SeqAllToAll5D:
    @staticmethod
    def apply(group, input, scatter_idx: int = 3, gather_idx: int = 1, use_sync: bool = False):
        return all_to_all_5D(input, scatter_idx, gather_idx, group=group, use_sync=use_sync)

    @staticmethod
    def forward(ctx, group, input, scatter_idx: int = 3, gather_idx: int = 1, use_sync: bool = False):
        return all_to_all_5D(input, scatter_idx, gather_idx, group=group, use_sync=use_sync)
