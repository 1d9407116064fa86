"""Model-specific extractors for hook-based caches — the contract of vllm_omni/diffusion/cache/teacache/extractors.py:24-268.

An extractor re-walks the transformer's forward through its PUBLIC modules (`img_in`, `txt_norm`, `txt_in`,
`time_text_embed`, `pos_embed`, `transformer_blocks[i]`, `.img_mod`, `.img_norm1`, `norm_out`, `proj_out`) and hands the cache
hook a `CacheContext`: the first block's modulated input (decision signal), a callable that runs the block stack and one that
finishes the forward.  On this transformer every one of those modules is a thin front of the C-ABI kernels (one block =
`omni_dit_block`), so hooks written against the reference's surface drive the MI355X kernels unchanged."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable

import torch
import torch.nn as nn


@dataclass
class CacheContext:
    modulated_input: torch.Tensor
    hidden_states: torch.Tensor
    encoder_hidden_states: torch.Tensor | None
    temb: torch.Tensor
    run_transformer_blocks: Callable[[], tuple[torch.Tensor, ...]]
    postprocess: Callable[[torch.Tensor], Any]
    extra_states: dict[str, Any] | None = None

    def validate(self) -> None:
        for name in ("modulated_input", "hidden_states", "temb"):
            if not isinstance(getattr(self, name), torch.Tensor):
                raise TypeError(f"{name} must be torch.Tensor, got {type(getattr(self, name))}")
        if not callable(self.run_transformer_blocks) or not callable(self.postprocess):
            raise TypeError("run_transformer_blocks and postprocess must be callable")
        if self.modulated_input.shape[0] != self.hidden_states.shape[0]:
            raise ValueError("Batch size mismatch between modulated_input and hidden_states")


def extract_qwen_context(module: nn.Module, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor,
                         encoder_hidden_states_mask: torch.Tensor = None, timestep: torch.Tensor = None,
                         img_shapes=None, txt_seq_lens=None, guidance=None, additional_t_cond=None,
                         attention_kwargs: dict | None = None, **kwargs: Any) -> CacheContext:
    """QwenImageTransformer2DModel (reference extractors.py:145-261; forward walk of qwen_image_transformer.py:743-798)."""
    if not hasattr(module, "transformer_blocks") or len(module.transformer_blocks) == 0:
        raise ValueError("Module must have transformer_blocks")
    if guidance is not None:
        raise NotImplementedError("guidance-embedding variants are outside the Qwen-Image T2I path")
    from ...models.qwen_image.qwen_image_transformer import Transformer2DModelOutput

    hidden = module.img_in(hidden_states)
    ts = timestep.to(device=hidden.device, dtype=hidden.dtype)
    enc = module.txt_in(module.txt_norm(encoder_hidden_states))
    temb = module.time_text_embed(ts, hidden, additional_t_cond)
    rotary = module.pos_embed(img_shapes, txt_seq_lens, device=hidden.device)
    first = module.transformer_blocks[0]
    img_mod1, _ = first.img_mod(temb).chunk(2, dim=-1)
    modulated, _ = first.img_norm1(hidden, img_mod1)

    def run_transformer_blocks():
        h, e = hidden, enc
        for block in module.transformer_blocks:
            e, h = block(hidden_states=h, encoder_hidden_states=e, encoder_hidden_states_mask=encoder_hidden_states_mask,
                         temb=temb, image_rotary_emb=rotary, joint_attention_kwargs=attention_kwargs)
        return (h, e)

    return_dict = kwargs.get("return_dict", True)

    def postprocess(h):
        out = module.proj_out(module.norm_out(h, temb))
        return Transformer2DModelOutput(out) if return_dict else (out,)

    return CacheContext(modulated_input=modulated, hidden_states=hidden, encoder_hidden_states=enc, temb=temb,
                        run_transformer_blocks=run_transformer_blocks, postprocess=postprocess)


EXTRACTOR_REGISTRY: dict[str, Callable] = {"QwenImageTransformer2DModel": extract_qwen_context}


def register_extractor(transformer_cls_name: str, fn: Callable) -> None:
    EXTRACTOR_REGISTRY[transformer_cls_name] = fn


def get_extractor(transformer_cls_name: str) -> Callable:
    if transformer_cls_name not in EXTRACTOR_REGISTRY:
        raise ValueError(f"Unknown model type: '{transformer_cls_name}'. Available: {list(EXTRACTOR_REGISTRY)}")
    return EXTRACTOR_REGISTRY[transformer_cls_name]
