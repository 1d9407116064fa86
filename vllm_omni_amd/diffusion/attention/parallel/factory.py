"""Strategy selection (vllm_omni/diffusion/attention/parallel/factory.py:12-45): Ulysses when the current diffusion config
asks for ulysses_degree > 1 and a process group is up, else the identity.  Ring attention is not inferred (as in the
reference)."""
from __future__ import annotations

import torch.distributed as dist

from ...data import get_current_omni_diffusion_config
from .base import NoParallelAttention
from .ulysses import UlyssesParallelAttention


def build_parallel_attention_strategy(*, scatter_idx: int = 2, gather_idx: int = 1, use_sync: bool = False, group=None):
    try:
        p = get_current_omni_diffusion_config().parallel_config
    except Exception:  # noqa: BLE001
        return NoParallelAttention()
    if getattr(p, "ulysses_degree", 1) > 1 and dist.is_initialized():
        return UlyssesParallelAttention(group=group, scatter_idx=scatter_idx, gather_idx=gather_idx, use_sync=use_sync)
    return NoParallelAttention()
