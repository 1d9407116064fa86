"""OmniRequestOutput — the diffusion half of vllm_omni/outputs.py:25-120 (`from_diffusion`): what `DiffusionEngine.step`
returns per request and what the OpenAI image endpoint serialises."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

import torch


@dataclass
class OmniRequestOutput:
    request_id: str = ""
    finished: bool = True
    stage_id: int | None = None
    final_output_type: str = "text"
    request_output: Any = None
    images: list = field(default_factory=list)          # PIL.Image.Image (or uint8 HWC arrays when PIL is not wanted)
    prompt: str | None = None
    latents: torch.Tensor | None = None
    metrics: dict[str, Any] = field(default_factory=dict)

    @classmethod
    def from_diffusion(cls, request_id: str, images: list, prompt: str | None = None, metrics: dict[str, Any] | None = None,
                       latents: torch.Tensor | None = None) -> "OmniRequestOutput":
        return cls(request_id=request_id, final_output_type="image", images=images, prompt=prompt, latents=latents,
                   metrics=metrics or {}, finished=True)
