from .base import NoParallelAttention, ParallelAttentionContext
from .factory import build_parallel_attention_strategy
from .ulysses import UlyssesParallelAttention

__all__ = ["NoParallelAttention", "ParallelAttentionContext", "UlyssesParallelAttention", "build_parallel_attention_strategy"]
