"""Attention plug-in contracts for the diffusion path.

API-compatible with the reference's `vllm_omni/diffusion/attention/backends/abstract.py:11-86` — a backend class
advertises a name, an implementation class and the head sizes it supports; an implementation is constructed per
attention layer and called with `[B, S, H, dh]` tensors plus an `AttentionMetadata` bag — but expressed as a small
registry-friendly base + a plain dataclass rather than a tower of abstract static methods.
"""
from __future__ import annotations

import abc
import dataclasses
from typing import ClassVar

import torch


@dataclasses.dataclass
class AttentionMetadata:
    """Per-call extras.  `joint_*` carry the text-stream q/k/v that a sequence-parallel caller keeps replicated;
    `joint_strategy` says on which side of the sequence they are concatenated."""
    attn_mask: torch.Tensor | None = None
    joint_query: torch.Tensor | None = None
    joint_key: torch.Tensor | None = None
    joint_value: torch.Tensor | None = None
    joint_strategy: str = "front"

    def __post_init__(self):
        if self.joint_strategy not in ("front", "rear"):
            raise ValueError(f"joint_strategy must be 'front' or 'rear', got {self.joint_strategy!r}")


class AttentionImpl(abc.ABC):
    """One instance per attention layer."""

    def __init__(self, num_heads: int, head_size: int, softmax_scale: float, causal: bool = False,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None:
        self.num_heads, self.head_size = num_heads, head_size
        self.softmax_scale, self.causal = softmax_scale, causal
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.prefix = prefix

    @abc.abstractmethod
    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                attn_metadata: AttentionMetadata | None = None) -> torch.Tensor:
        """q, k, v: [B, S, H, dh] -> [B, S, H, dh]."""


class AttentionBackend(abc.ABC):
    """Static description of a backend; the selector returns the class, never an instance."""
    accept_output_buffer: ClassVar[bool] = False
    NAME: ClassVar[str] = ""
    IMPL: ClassVar[type[AttentionImpl] | None] = None
    HEAD_SIZES: ClassVar[tuple[int, ...]] = ()      # empty = any

    @classmethod
    def get_name(cls) -> str:
        return cls.NAME

    @classmethod
    def get_impl_cls(cls) -> type[AttentionImpl]:
        if cls.IMPL is None:
            raise NotImplementedError(f"{cls.__name__} declares no implementation class")
        return cls.IMPL

    @classmethod
    def get_metadata_cls(cls) -> type[AttentionMetadata]:
        return AttentionMetadata

    @classmethod
    def get_builder_cls(cls):
        """The reference declares a metadata-builder hook (abstract.py:31-34) that nothing on the diffusion path calls; a
        backend has none unless it says so."""
        return None

    @classmethod
    def get_supported_head_sizes(cls) -> list[int]:
        return list(cls.HEAD_SIZES)

    @classmethod
    def supports_head_size(cls, head_size: int) -> bool:
        return not cls.HEAD_SIZES or head_size in cls.HEAD_SIZES
