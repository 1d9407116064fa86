"""Config / output dataclasses — the subset of vllm_omni/diffusion/data.py the DiT path touches
(DiffusionParallelConfig :26-91, TransformerConfig :95-117, OmniDiffusionConfig :237-455, DiffusionOutput :508-520).
Field names and defaults follow the reference so configs written for it construct here unchanged."""
from __future__ import annotations

from collections.abc import Callable
from contextlib import contextmanager
from dataclasses import dataclass, field
from typing import Any

import torch


@dataclass
class DiffusionParallelConfig:
    pipeline_parallel_size: int = 1
    data_parallel_size: int = 1
    tensor_parallel_size: int = 1
    sequence_parallel_size: int | None = None
    ulysses_degree: int = 1
    ring_degree: int = 1
    cfg_parallel_size: int = 1

    def __post_init__(self) -> None:
        if self.sequence_parallel_size is None:
            self.sequence_parallel_size = self.ulysses_degree * self.ring_degree
        for n in ("pipeline_parallel_size", "data_parallel_size", "tensor_parallel_size", "sequence_parallel_size",
                  "ulysses_degree", "ring_degree", "cfg_parallel_size"):
            assert getattr(self, n) > 0, f"{n} must be > 0"
        assert self.sequence_parallel_size == self.ulysses_degree * self.ring_degree
        # this build shards by REQUEST (data parallel): everything else must stay 1
        for n in ("pipeline_parallel_size", "tensor_parallel_size", "sequence_parallel_size", "cfg_parallel_size"):
            if getattr(self, n) != 1:
                raise NotImplementedError(f"{n} > 1 is not built; the MI355X path is data-parallel over requests")
        self.world_size = self.data_parallel_size

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "DiffusionParallelConfig":
        if not isinstance(data, dict):
            raise TypeError(f"Expected parallel config dict, got {type(data)!r}")
        return cls(**data)


@dataclass
class TransformerConfig:
    params: dict[str, Any] = field(default_factory=dict)

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "TransformerConfig":
        if not isinstance(data, dict):
            raise TypeError(f"Expected transformer config dict, got {type(data)!r}")
        return cls(params=dict(data))

    def to_dict(self) -> dict[str, Any]:
        return dict(self.params)

    def get(self, key: str, default: Any | None = None) -> Any:
        return self.params.get(key, default)

    def __getattr__(self, item: str) -> Any:
        params = object.__getattribute__(self, "params")
        try:
            return params[item]
        except KeyError as exc:
            raise AttributeError(item) from exc


@dataclass
class OmniDiffusionConfig:
    model: str | None = None
    model_class_name: str | None = "QwenImagePipeline"
    dtype: torch.dtype = torch.bfloat16
    tf_model_config: TransformerConfig = field(default_factory=TransformerConfig)
    parallel_config: DiffusionParallelConfig = field(default_factory=DiffusionParallelConfig)
    cache_backend: str = "none"
    cache_config: dict[str, Any] = field(default_factory=dict)
    num_gpus: int | None = None
    master_port: int | None = None
    output_type: str = "pil"
    max_step_batch: int = 4          # NEW (not in the reference): requests whose steps share one DiT forward
    dist_timeout: int | None = None

    def __post_init__(self):
        if isinstance(self.parallel_config, dict):
            self.parallel_config = DiffusionParallelConfig.from_dict(self.parallel_config)
        if isinstance(self.tf_model_config, dict):
            self.tf_model_config = TransformerConfig.from_dict(self.tf_model_config)
        if self.num_gpus is None:
            self.num_gpus = self.parallel_config.world_size
        if self.cache_backend not in ("none", None):
            raise NotImplementedError("cache backends (TeaCache / cache-dit) are SURVEY.md §8f row N3")


_current: OmniDiffusionConfig | None = None


@contextmanager
def set_current_omni_diffusion_config(cfg: OmniDiffusionConfig, check_compile=False, prefix: str | None = None):
    global _current
    old = _current
    _current = cfg
    try:
        yield
    finally:
        _current = old


def get_current_omni_diffusion_config() -> OmniDiffusionConfig:
    return _current if _current is not None else OmniDiffusionConfig()


@dataclass
class DiffusionOutput:
    output: torch.Tensor | None = None
    trajectory_timesteps: list[torch.Tensor] | None = None
    trajectory_latents: torch.Tensor | None = None
    trajectory_decoded: list[torch.Tensor] | None = None
    error: str | None = None
    post_process_func: Callable[..., Any] | None = None
