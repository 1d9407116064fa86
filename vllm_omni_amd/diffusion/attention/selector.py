"""Backend selection — mirror of vllm_omni/diffusion/attention/selector.py:18-77.

Same env var (DIFFUSION_ATTENTION_BACKEND) and the same `{name: {module, class}}` registry shape; CDNA4_FLASH is the default
wherever a GPU is visible; TORCH_SDPA (the reference's default) serves CPU-only hosts and explicit requests.  Unknown names raise ValueError like the reference.
"""
import importlib
import os
from functools import cache

from .backends.abstract import AttentionBackend

_BACKEND_CONFIG = {
    "CDNA4_FLASH": {"module": "vllm_omni_amd.diffusion.attention.backends.cdna4_flash", "class": "CDNA4FlashBackend"},
    "TORCH_SDPA": {"module": "vllm_omni_amd.diffusion.attention.backends.sdpa", "class": "SDPABackend"},
}


def load_backend(backend_name: str) -> type[AttentionBackend]:
    cfg = _BACKEND_CONFIG[backend_name]
    return getattr(importlib.import_module(cfg["module"]), cfg["class"])


@cache
def get_attn_backend(head_size: int) -> type[AttentionBackend]:
    name = os.environ.get("DIFFUSION_ATTENTION_BACKEND")
    if name is not None:
        up = name.upper()
        if up not in _BACKEND_CONFIG:
            raise ValueError(f"Invalid attention backend for diffusion: '{name}'. Valid backends are: "
                             f"{list(_BACKEND_CONFIG)}")
        if up == "TORCH_SDPA":
            import torch

            if torch.cuda.is_available():
                # refused HERE, not at the first forward (round-5 advisor): on a GPU host the reference's default backend would
                # run F.scaled_dot_product_attention on device tensors — this build has no torch attention on the device (no dual
                # backend); CDNA4_FLASH honours what SDPAImpl honours (attn_mask, cross-attention, causal, head sizes 64 / 128)
                raise ValueError("DIFFUSION_ATTENTION_BACKEND=TORCH_SDPA serves CPU-only hosts; a GPU is visible here: unset the "
                                 "variable (CDNA4_FLASH takes attn_mask / cross-attention / causal / head sizes 64 and 128)")
        return load_backend(up)
    import torch

    # default: the HIP kernel wherever a GPU is visible; a CPU-only host (host-logic tests) gets the reference's SDPA default
    return load_backend("CDNA4_FLASH" if torch.cuda.is_available() else "TORCH_SDPA")
