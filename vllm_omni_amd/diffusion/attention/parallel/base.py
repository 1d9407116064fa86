"""Parallel-attention strategy contract (vllm_omni/diffusion/attention/parallel/base.py:14-84): orthogonal to the kernel
backend — `pre_attention` reshards q/k/v before the kernel, `post_attention` reshards its output back."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ParallelAttentionContext:
    name: str


class NoParallelAttention:
    enabled = False
    name = "none"

    def pre_attention(self, query, key, value, attn_metadata):
        return query, key, value, attn_metadata, None

    def post_attention(self, attn_output, ctx):
        return attn_output
