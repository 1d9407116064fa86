"""OmniDiffusionRequest — the fields of vllm_omni/diffusion/request.py:14-187 the Qwen-Image T2I path reads
(pipeline_qwen_image.py:614-624), plus pre-computed embeddings (the text encoder is SURVEY.md §8f row N1)."""
from __future__ import annotations

from dataclasses import dataclass, field

import torch


@dataclass
class OmniDiffusionRequest:
    prompt: str | list[str] | None = None
    negative_prompt: str | list[str] | None = None
    request_id: str | None = None
    height: int | None = None
    width: int | None = None
    num_inference_steps: int = 50
    true_cfg_scale: float | None = None
    guidance_scale: float = 1.0
    num_outputs_per_prompt: int = 1
    seed: int | None = None
    generator: torch.Generator | None = None
    latents: torch.Tensor | None = None               # packed [1, S_img, 64]; injected for parity runs
    prompt_embeds: torch.Tensor | None = None          # [1, T, 3584]
    prompt_embeds_mask: torch.Tensor | None = None
    negative_prompt_embeds: torch.Tensor | None = None
    negative_prompt_embeds_mask: torch.Tensor | None = None
    output_type: str = "pt"
    extra: dict = field(default_factory=dict)
