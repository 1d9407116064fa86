"""Architecture name -> pipeline class, and the per-pipeline pre/post-process hooks — the role of
vllm_omni/diffusion/registry.py:10-146 for the pipelines this build carries.  `model_class_name` comes from a checkpoint's
`model_index.json` (`_class_name`) exactly as in the reference (entrypoints/omni_diffusion.py:51-55)."""
from __future__ import annotations

import importlib

from .data import OmniDiffusionConfig

_PKG = "vllm_omni_amd.diffusion.models"
# arch name: (folder, module, class)
_DIFFUSION_MODELS = {
    "QwenImagePipeline": ("qwen_image", "pipeline_qwen_image", "QwenImagePipeline"),
    "QwenImageEditPipeline": ("qwen_image", "pipeline_qwen_image_edit", "QwenImageEditPipeline"),
    "QwenImageEditPlusPipeline": ("qwen_image", "pipeline_qwen_image_edit_plus", "QwenImageEditPlusPipeline"),
    "QwenImageLayeredPipeline": ("qwen_image", "pipeline_qwen_image_layered", "QwenImageLayeredPipeline"),
}
_POST_PROCESS = {"QwenImagePipeline": "get_qwen_image_post_process_func",
                 "QwenImageEditPipeline": "get_qwen_image_post_process_func",
                 "QwenImageEditPlusPipeline": "get_qwen_image_post_process_func",
                 "QwenImageLayeredPipeline": "get_qwen_image_post_process_func"}
_PRE_PROCESS: dict[str, str] = {"QwenImageLayeredPipeline": "get_qwen_image_layered_pre_process_func"}


def _module(arch: str):
    if arch not in _DIFFUSION_MODELS:
        raise ValueError(f"Model class {arch} not found in diffusion model registry. Known: {sorted(_DIFFUSION_MODELS)}")
    folder, mod, _ = _DIFFUSION_MODELS[arch]
    return importlib.import_module(f"{_PKG}.{folder}.{mod}")


def resolve_model_cls(arch: str):
    return getattr(_module(arch), _DIFFUSION_MODELS[arch][2])


def apply_vae_memory_flags(model, od_config: OmniDiffusionConfig):
    """od_config.vae_use_slicing / vae_use_tiling -> the pipeline's VAE (reference registry.py:88-92)."""
    vae = getattr(model, "vae", None)
    if vae is not None:
        if hasattr(vae, "use_slicing"):
            vae.use_slicing = bool(getattr(od_config, "vae_use_slicing", False))
        if hasattr(vae, "use_tiling"):
            vae.use_tiling = bool(getattr(od_config, "vae_use_tiling", False))
    return model


def initialize_model(od_config: OmniDiffusionConfig, **kw):
    """Instantiate `od_config.model_class_name` with the registry contract `__init__(*, od_config, prefix="")`, then hand the
    VAE its memory flags."""
    return apply_vae_memory_flags(resolve_model_cls(od_config.model_class_name or "QwenImagePipeline")(od_config=od_config, **kw),
                                  od_config)


def _load_process_func(od_config: OmniDiffusionConfig, table: dict[str, str]):
    arch = od_config.model_class_name or "QwenImagePipeline"
    if arch not in table:
        return None
    mod = _module(arch)
    name = table[arch]
    fn = getattr(mod, name, None)
    if fn is None:                       # the Edit module re-uses the T2I post-processing
        fn = getattr(importlib.import_module(f"{_PKG}.qwen_image.pipeline_qwen_image"), name)
    return fn(od_config)


def get_diffusion_post_process_func(od_config: OmniDiffusionConfig):
    return _load_process_func(od_config, _POST_PROCESS)


def get_diffusion_pre_process_func(od_config: OmniDiffusionConfig):
    return _load_process_func(od_config, _PRE_PROCESS)
