"""CustomOp — the per-platform forward dispatch used by the diffusion layers.

Same observable behaviour as the reference base class (`vllm_omni/diffusion/layers/custom_op.py:9-49`): a subclass
implements `forward_<platform>` methods, the choice is made once at construction and `forward()` calls it.  This
build knows two execution targets — `hip` (PyTorch-ROCm on a gfx950 GPU) and `native` (plain PyTorch, CPU tensors /
tests) — and keeps `forward_cuda` / `forward_npu` only so that reference subclasses that override them still import.
"""
from __future__ import annotations

from typing import Any, Callable

import torch
import torch.nn as nn


def is_rocm() -> bool:
    return torch.version.hip is not None and torch.cuda.is_available()


def current_target() -> str:
    return "hip" if is_rocm() else "native"


class CustomOp(nn.Module):
    _TARGET_METHOD = {"hip": "forward_hip", "native": "forward_native"}

    def __init__(self) -> None:
        super().__init__()
        self._forward_method: Callable[..., Any] = self.dispatch_forward()

    def dispatch_forward(self) -> Callable[..., Any]:
        return getattr(self, self._TARGET_METHOD[current_target()])

    def forward(self, *args, **kwargs) -> Any:
        return self._forward_method(*args, **kwargs)

    # --- targets --------------------------------------------------------------------------------------------------
    def forward_hip(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} has no HIP kernel")

    def forward_native(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} has no PyTorch statement")

    def forward_cuda(self, *args, **kwargs):
        raise NotImplementedError("gfx950-only build: there is no CUDA path")

    def forward_npu(self, *args, **kwargs):
        raise NotImplementedError("gfx950-only build: there is no NPU path")
