"""CustomOp dispatch base — mirror of vllm_omni/diffusion/layers/custom_op.py:9-49.

Same contract: subclasses provide forward_hip / forward_native (and optionally forward_cuda / forward_npu);
`forward` dispatches once at construction.  On this platform (PyTorch-ROCm on MI355X) the dispatch target is
`forward_hip`; `forward_native` is the plain-PyTorch statement kept for CPU tensors and tests.
"""
from collections.abc import Callable
from typing import Any

import torch
import torch.nn as nn


def is_rocm() -> bool:
    return torch.version.hip is not None and torch.cuda.is_available()


class CustomOp(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self._forward_method = self.dispatch_forward()

    def dispatch_forward(self) -> Callable:
        return self.forward_hip if is_rocm() else self.forward_native

    def forward(self, *args, **kwargs) -> Any:
        return self._forward_method(*args, **kwargs)

    def forward_native(self, *args, **kwargs):
        raise NotImplementedError

    def forward_cuda(self, *args, **kwargs):
        raise NotImplementedError("this build targets gfx950 only; there is no CUDA path")

    def forward_npu(self, *args, **kwargs):
        raise NotImplementedError("this build targets gfx950 only; there is no NPU path")

    def forward_hip(self, *args, **kwargs):
        raise NotImplementedError
