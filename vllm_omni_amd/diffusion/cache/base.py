"""CacheBackend contract — mirror of vllm_omni/diffusion/cache/base.py:31-105: a backend is built from the cache config,
`enable(pipeline)` is called once after the pipeline is loaded (gpu_worker.py:104-107) and `refresh(pipeline,
num_inference_steps)` before every generation (gpu_worker.py:132-134)."""
from __future__ import annotations

import abc
from typing import Any


class CacheBackend(abc.ABC):
    def __init__(self, config: Any):
        self.config = config
        self.enabled = False

    @abc.abstractmethod
    def enable(self, pipeline: Any) -> None:
        ...

    @abc.abstractmethod
    def refresh(self, pipeline: Any, num_inference_steps: int, verbose: bool = True) -> None:
        ...

    def is_enabled(self) -> bool:
        return self.enabled
