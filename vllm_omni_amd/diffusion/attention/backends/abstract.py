"""Attention backend interfaces — mirror of vllm_omni/diffusion/attention/backends/abstract.py:11-86."""
from abc import ABC, abstractmethod
from dataclasses import dataclass

import torch


class AttentionBackend(ABC):
    accept_output_buffer: bool = False

    @staticmethod
    @abstractmethod
    def get_name() -> str: ...

    @staticmethod
    @abstractmethod
    def get_impl_cls(): ...

    @staticmethod
    @abstractmethod
    def get_supported_head_sizes() -> list[int]: ...

    @classmethod
    def supports_head_size(cls, head_size: int) -> bool:
        sizes = cls.get_supported_head_sizes()
        return (not sizes) or head_size in sizes


@dataclass
class AttentionMetadata:
    attn_mask: torch.Tensor | None = None
    joint_query: torch.Tensor | None = None
    joint_key: torch.Tensor | None = None
    joint_value: torch.Tensor | None = None
    joint_strategy: str = "front"


class AttentionImpl(ABC):
    @abstractmethod
    def __init__(self, num_heads: int, head_size: int, softmax_scale: float, causal: bool = False,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None: ...

    @abstractmethod
    def forward(self, query, key, value, attn_metadata=None) -> torch.Tensor: ...
