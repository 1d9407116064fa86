"""OmniRequestOutput — vllm_omni/outputs.py:25-170 as the diffusion path uses it: what `DiffusionEngine.step` returns per request
and what the OpenAI image endpoint serialises (`from_diffusion`, `num_images`, `to_dict`); `from_pipeline` wraps a stage's own
output object unchanged (the LLM stages are the reference's: nothing of theirs is re-implemented here)."""
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

    @classmethod
    def from_pipeline(cls, stage_id: int, final_output_type: str, request_output: Any) -> "OmniRequestOutput":
        return cls(request_id=getattr(request_output, "request_id", ""), stage_id=stage_id, final_output_type=final_output_type,
                   request_output=request_output, finished=True)

    @property
    def num_images(self) -> int:
        return len(self.images)

    @property
    def is_diffusion_output(self) -> bool:
        return len(self.images) > 0 or self.final_output_type == "image"

    @property
    def is_pipeline_output(self) -> bool:
        return self.stage_id is not None and self.request_output is not None

    def to_dict(self) -> dict[str, Any]:
        """JSON-serialisable summary (pictures are counted, not embedded)."""
        out: dict[str, Any] = {"request_id": self.request_id, "finished": self.finished, "final_output_type": self.final_output_type}
        if self.is_diffusion_output:
            out.update(num_images=self.num_images, prompt=self.prompt, metrics=self.metrics)
        if self.is_pipeline_output:
            out.update(stage_id=self.stage_id)
        return out

    def __repr__(self) -> str:
        return (f"OmniRequestOutput(request_id={self.request_id!r}, finished={self.finished}, final_output_type="
                f"{self.final_output_type!r}, images=[{len(self.images)} images], prompt={self.prompt!r}, "
                f"latents={None if self.latents is None else tuple(self.latents.shape)}, metrics={self.metrics})")
