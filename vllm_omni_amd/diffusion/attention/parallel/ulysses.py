"""Ulysses sequence-parallel attention strategy — the semantics of vllm_omni/diffusion/attention/parallel/ulysses.py:27-135:
every rank holds seq/P tokens and all heads; an all-to-all turns that into all tokens and heads/P around the kernel.
`AttentionMetadata.joint_*` (text tokens, replicated on every rank): joint_query is concatenated to the query BEFORE the
all-to-all, joint_key/value are head-sliced for this rank and concatenated AFTER it ("front" or "rear")."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any

import torch
import torch.distributed as dist

from ...distributed.comm import SeqAllToAll4D
from .base import ParallelAttentionContext


@dataclass(frozen=True)
class _UlyssesCtx(ParallelAttentionContext):
    group: Any = None
    scatter_idx: int = 2
    gather_idx: int = 1
    use_sync: bool = False


class UlyssesParallelAttention:
    enabled = True
    name = "ulysses"

    def __init__(self, group=None, scatter_idx: int = 2, gather_idx: int = 1, use_sync: bool = False, sp_group=None):
        # `sp_group` (reference SequenceParallelGroupCoordinator) is accepted for signature compatibility
        self._pg = getattr(sp_group, "ulysses_group", None) if sp_group is not None else group
        self._scatter_idx, self._gather_idx, self._use_sync = scatter_idx, gather_idx, use_sync

    def _world_rank(self):
        if not dist.is_initialized():
            return 1, 0
        return dist.get_world_size(self._pg), dist.get_rank(self._pg)

    def pre_attention(self, query, key, value, attn_metadata):
        jq = jk = jv = strategy = None
        if attn_metadata is not None:
            jq, jk, jv, strategy = (attn_metadata.joint_query, attn_metadata.joint_key, attn_metadata.joint_value,
                                    attn_metadata.joint_strategy)
        given = [t is not None for t in (jq, jk, jv)]
        if any(given) and not all(given):
            raise ValueError("joint_query, joint_key, and joint_value should be None or not None simultaneously.")
        joint = all(given)
        if joint:
            if strategy not in ("front", "rear"):
                raise ValueError(f"joint_strategy: {strategy} not supported. supported joint strategy: ['front', 'rear']")
            query = torch.cat([query, jq], dim=1) if strategy == "rear" else torch.cat([jq, query], dim=1)
            P, r = self._world_rank()
            hp = jk.shape[-2] // P
            jk, jv = jk[..., hp * r: hp * (r + 1), :], jv[..., hp * r: hp * (r + 1), :]
        a2a = lambda t: SeqAllToAll4D.apply(self._pg, t, self._scatter_idx, self._gather_idx, self._use_sync)  # noqa: E731
        query, key, value = a2a(query), a2a(key), a2a(value)
        if joint:
            key = torch.cat([jk, key], dim=1) if strategy == "front" else torch.cat([key, jk], dim=1)
            value = torch.cat([jv, value], dim=1) if strategy == "front" else torch.cat([value, jv], dim=1)
            import dataclasses

            attn_metadata = dataclasses.replace(attn_metadata, joint_query=None, joint_key=None, joint_value=None)
        return query, key, value, attn_metadata, _UlyssesCtx(name=self.name, group=self._pg, scatter_idx=self._scatter_idx,
                                                             gather_idx=self._gather_idx, use_sync=self._use_sync)

    def post_attention(self, attn_output, ctx):
        if not isinstance(ctx, _UlyssesCtx):
            raise TypeError(f"Unexpected ctx type: {type(ctx)!r}")
        return SeqAllToAll4D.apply(ctx.group, attn_output, ctx.gather_idx, ctx.scatter_idx, ctx.use_sync)
