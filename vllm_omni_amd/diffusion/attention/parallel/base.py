"""Parallel-attention strategy contract (vllm_omni/diffusion/attention/parallel/base.py:14-84): orthogonal to the kernel
backend — `pre_attention` reshards q/k/v before the kernel, `post_attention` reshards its output back."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Protocol, runtime_checkable


@dataclass(frozen=True)
class ParallelAttentionContext:
    name: str


@runtime_checkable
class ParallelAttentionStrategy(Protocol):
    """What a strategy object offers (typing contract; `UlyssesParallelAttention` and `NoParallelAttention` satisfy it)."""
    enabled: bool
    name: str

    def pre_attention(self, query: Any, key: Any, value: Any, attn_metadata: Any) -> tuple: ...

    def post_attention(self, attn_output: Any, ctx: Any) -> Any: ...


class NoParallelAttention:
    enabled = False
    name = "none"

    def pre_attention(self, query, key, value, attn_metadata):
        return query, key, value, attn_metadata, None

    def post_attention(self, attn_output, ctx):
        return attn_output
