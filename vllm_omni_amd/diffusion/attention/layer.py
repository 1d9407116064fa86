"""Attention layer — mirror of vllm_omni/diffusion/attention/layer.py:17-70 (backend impl between the
parallel strategy's pre/post hooks).  Data-parallel serving needs no resharding (identity strategy,
`NoParallelAttention`); with `ulysses_degree > 1` the Ulysses strategy (parallel/ulysses.py) wraps the kernel in all-to-alls."""
import torch
import torch.nn as nn

from .backends.abstract import AttentionMetadata
from .parallel import build_parallel_attention_strategy
from .selector import get_attn_backend


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_size: int, causal: bool, softmax_scale: float,
                 num_kv_heads: int | None = None, prefix: str = "", scatter_idx: int = 2, gather_idx: int = 1,
                 use_sync: bool = False):
        super().__init__()
        self.attn_backend = get_attn_backend(-1)
        self.attention = self.attn_backend.get_impl_cls()(num_heads=num_heads, head_size=head_size,
                                                          softmax_scale=softmax_scale, causal=causal,
                                                          num_kv_heads=num_kv_heads)
        self.softmax_scale = softmax_scale
        # sharding / communication around the kernel is a separate, pluggable strategy (reference layer.py:41-52)
        self.parallel_strategy = build_parallel_attention_strategy(scatter_idx=scatter_idx, gather_idx=gather_idx,
                                                                   use_sync=use_sync)

    def forward(self, query, key, value, attn_metadata: AttentionMetadata = None) -> torch.Tensor:
        if self.parallel_strategy.enabled:
            query, key, value, attn_metadata, ctx = self.parallel_strategy.pre_attention(query, key, value, attn_metadata)
            out = self.attention.forward(query, key, value, attn_metadata)
            out = out[0] if isinstance(out, tuple) else out
            return self.parallel_strategy.post_attention(out, ctx)
        if attn_metadata is not None and attn_metadata.joint_query is not None:
            # SP-style call: the replicated text q/k/v ride in the metadata (reference ulysses.py:83-121)
            front = attn_metadata.joint_strategy == "front"
            cat = (lambda j, x: torch.cat([j, x], 1)) if front else (lambda j, x: torch.cat([x, j], 1))
            query, key, value = cat(attn_metadata.joint_query, query), cat(attn_metadata.joint_key, key), \
                cat(attn_metadata.joint_value, value)
            # the joint tensors are folded in; what remains of the metadata (e.g. an attn_mask) still reaches the impl,
            # which rejects what it cannot honour instead of silently attending unmasked
            attn_metadata = AttentionMetadata(attn_mask=attn_metadata.attn_mask, joint_strategy=attn_metadata.joint_strategy)
        out = self.attention.forward(query, key, value, attn_metadata)
        return out[0] if isinstance(out, tuple) else out
