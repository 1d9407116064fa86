"""OmniDiffusionRequest — the fields of vllm_omni/diffusion/request.py:14-187 the Qwen-Image pipelines read
(pipeline_qwen_image.py:614-624; the Edit / Edit-Plus / Layered pipelines' `pil_image`, `preprocessed_image`, `prompt_image`,
`layers`, `resolution`, `cfg_normalize`, `use_en_prompt`: pipeline_qwen_image_edit.py:64-93,671-674,
pipeline_qwen_image_layered.py:69-100,668-685), plus pre-computed embeddings (the text encoder is SURVEY.md §8f row N1).

The pipelines of this build read pictures and variant knobs from `extra` (`image`, `prompt_image`, `image_latents`, `layers`,
`resolution`, `cfg_normalize`, `use_en_prompt`); a caller written against the reference sets the reference's first-class fields
instead — `__post_init__` folds them into `extra` (an explicit `extra` entry wins), so both spellings reach the same code."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

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
    generator: torch.Generator | list[torch.Generator] | None = None
    latents: torch.Tensor | None = None               # packed [1, S_img, 64]; injected for parity runs
    prompt_embeds: torch.Tensor | None = None          # [1, T, 3584] (the reference's default, an empty list, means "none")
    prompt_embeds_mask: torch.Tensor | None = None
    negative_prompt_embeds: torch.Tensor | None = None
    negative_prompt_embeds_mask: torch.Tensor | None = None
    output_type: str = "pt"
    extra: dict = field(default_factory=dict)
    # ---- the reference's spellings of the image / variant inputs (request.py:33-38,70-79) ----
    pil_image: Any = None                              # PIL image / tensor; a list of them for Edit-Plus  -> extra["image"]
    preprocessed_image: torch.Tensor | None = None     # what the reference's pre-process step leaves here  -> extra["image"]
    prompt_image: Any = None                           # the picture(s) the vision tower sees                -> extra["prompt_image"]
    layers: int | None = None                          # Layered: number of output layers (reference default 4)
    resolution: int | None = None                      # Layered: 640 or 1024 bucket
    cfg_normalize: bool | None = None                  # Layered: norm-rescaled true-CFG (reference default False)
    use_en_prompt: bool | None = None                  # Layered: caption language
    max_sequence_length: int | None = None

    def __post_init__(self) -> None:
        if isinstance(self.prompt_embeds, (list, tuple)):          # reference type: list[Tensor] | Tensor, default []
            self.prompt_embeds = None if len(self.prompt_embeds) == 0 else (
                self.prompt_embeds[0] if len(self.prompt_embeds) == 1 else torch.stack(list(self.prompt_embeds)))
        if isinstance(self.negative_prompt_embeds, (list, tuple)):
            self.negative_prompt_embeds = None if len(self.negative_prompt_embeds) == 0 else (
                self.negative_prompt_embeds[0] if len(self.negative_prompt_embeds) == 1
                else torch.stack(list(self.negative_prompt_embeds)))
        if self.extra is None:
            self.extra = {}
        picture = self.preprocessed_image if self.preprocessed_image is not None else self.pil_image
        for key, val in (("image", picture), ("prompt_image", self.prompt_image), ("layers", self.layers),
                         ("resolution", self.resolution), ("cfg_normalize", self.cfg_normalize),
                         ("use_en_prompt", self.use_en_prompt)):
            if val is not None and self.extra.get(key) is None:
                self.extra[key] = val
