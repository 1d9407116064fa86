"""Loads a diffusers-layout checkpoint DIRECTORY into a pipeline of this build — the role of
vllm_omni/diffusion/model_loader/diffusers_loader.py:35-260 for local paths.

Same surface: `DiffusersPipelineLoader.ComponentSource(model_or_path, subfolder, revision, prefix, fall_back_to_pt,
allow_patterns_overrides)`, `get_all_weights(model)` walking `model.weights_sources`, and `load_model(od_config,
load_device)` = registry `initialize_model` -> `model.load_weights(all weights)` -> every parameter must have been
loaded (reference :233-260).  Differences, all at the edges of the hot path:
  * no hub download (this build has no network dependency): a model id that is not a directory raises;
  * the safetensors files are read with `safetensors.safe_open` directly (the reference goes through vLLM's
    `safetensors_weights_iterator`); `*.bin` / `*.pt` fall back to `torch.load(weights_only=True)`;
  * when `diffusion_pytorch_model.safetensors.index.json` exists only the shards its `weight_map` names are read
    (the reference's `filter_duplicate_safetensors_files`, :139-147);
  * the reference builds its VAE with diffusers' `from_pretrained(subfolder="vae")`; here the pipelines list the VAE as a
    second `ComponentSource` (prefix "vae.") and `scheduler/scheduler_config.json` is applied to the Flow-Match schedule.
"""
from __future__ import annotations

import dataclasses
import glob
import json
import os
from collections.abc import Generator, Iterable

import torch

from ..data import OmniDiffusionConfig

MODEL_INDEX = "model_index.json"
DIFFUSION_MODEL_WEIGHTS_INDEX = "diffusion_pytorch_model.safetensors.index.json"


class DiffusersPipelineLoader:
    @dataclasses.dataclass
    class ComponentSource:
        model_or_path: str
        subfolder: str | None
        revision: str | None = None
        prefix: str = ""
        fall_back_to_pt: bool = True
        allow_patterns_overrides: list[str] | None = None

    def __init__(self, load_config=None):
        self.load_config = load_config

    # ------------------------------------------------------------------ files of one component
    def _prepare_weights(self, model_name_or_path: str, subfolder: str | None, revision: str | None = None,
                         fall_back_to_pt: bool = True,
                         allow_patterns_overrides: list[str] | None = None) -> tuple[str, list[str], bool]:
        if not os.path.isdir(model_name_or_path):
            raise FileNotFoundError(f"{model_name_or_path!r} is not a local checkpoint directory (this build does not "
                                    "download from the hub; pass the path of a diffusers-layout checkpoint)")
        patterns = ["*.safetensors", "*.bin"] + (["*.pt"] if fall_back_to_pt else [])
        if allow_patterns_overrides is not None:
            patterns = list(allow_patterns_overrides)
        folder = os.path.join(model_name_or_path, subfolder) if subfolder else model_name_or_path
        files: list[str] = []
        use_safetensors = False
        for pattern in patterns:
            files = sorted(glob.glob(os.path.join(folder, pattern)))
            if files:
                use_safetensors = pattern.endswith(".safetensors")
                break
        if use_safetensors:
            index = os.path.join(folder, DIFFUSION_MODEL_WEIGHTS_INDEX)
            if os.path.isfile(index):                       # keep only the shards the index names
                with open(index) as fh:
                    named = set(json.load(fh).get("weight_map", {}).values())
                files = [f for f in files if os.path.basename(f) in named] or files
        if not files:
            raise RuntimeError(f"Cannot find any model weights with `{folder}`")
        return folder, files, use_safetensors

    @staticmethod
    def _iterate(files: list[str], use_safetensors: bool) -> Generator[tuple[str, torch.Tensor], None, None]:
        if use_safetensors:
            from safetensors import safe_open

            for f in files:
                with safe_open(f, framework="pt", device="cpu") as fh:
                    for name in fh.keys():
                        yield name, fh.get_tensor(name)
        else:
            for f in files:
                state = torch.load(f, map_location="cpu", weights_only=True)
                yield from state.items()

    def _get_weights_iterator(self, source: "DiffusersPipelineLoader.ComponentSource"):
        _, files, use_safetensors = self._prepare_weights(source.model_or_path, source.subfolder, source.revision,
                                                          source.fall_back_to_pt, source.allow_patterns_overrides)
        return ((source.prefix + name, t) for name, t in self._iterate(files, use_safetensors))

    def get_all_weights(self, model) -> Generator[tuple[str, torch.Tensor], None, None]:
        sources: Iterable[DiffusersPipelineLoader.ComponentSource] = getattr(model, "weights_sources", ())
        for source in sources:
            yield from self._get_weights_iterator(source)

    # ------------------------------------------------------------------ whole pipeline
    def load_model(self, od_config: OmniDiffusionConfig, load_device: str | torch.device = "cuda", **model_kwargs):
        """Instantiate `od_config.model_class_name` and fill it from the checkpoint directory `od_config.model`."""
        from ..registry import initialize_model

        model = initialize_model(od_config, device=torch.device(load_device), **model_kwargs)
        if not getattr(model, "weights_sources", None):
            model.weights_sources = default_weight_sources(od_config.model)
        expected = model.expected_weight_names() if hasattr(model, "expected_weight_names") \
            else {n for n, _ in model.named_parameters()}
        loaded = model.load_weights(self.get_all_weights(model))
        missing = sorted(expected - set(loaded))
        if missing:
            raise ValueError(f"Following weights were not initialized from checkpoint: {missing[:8]}"
                             f"{' ...' if len(missing) > 8 else ''}")
        sched = os.path.join(od_config.model, "scheduler", "scheduler_config.json")
        if os.path.isfile(sched) and hasattr(model, "scheduler"):
            with open(sched) as fh:
                apply_scheduler_config(model.scheduler, json.load(fh))
        # prompt encoder: built when the checkpoint carries one (the reference always loads it, pipeline_qwen_image.py:
        # 225-228); without it requests must bring prompt_embeds
        if hasattr(model, "load_text_encoder") and getattr(model, "text_encoder", None) is None and \
                os.path.isdir(os.path.join(od_config.model, "text_encoder")) and \
                os.path.isdir(os.path.join(od_config.model, "tokenizer")):
            # a checkpoint directory that lacks a piece the encoder needs (an Edit checkpoint without processor/, a trimmed
            # local copy, a transformers build without the VL classes) still serves requests that bring prompt_embeds — as it
            # did before the auto-load existed; `od_config.cache_config`-style switch: extra key load_text_encoder=False skips it
            if getattr(od_config, "load_text_encoder", True):
                try:
                    model.load_text_encoder(od_config.model, device=load_device)
                except (FileNotFoundError, ImportError, OSError) as e:
                    print(f"[DiffusersPipelineLoader] prompt encoder not loaded ({type(e).__name__}: {e}); "
                          "requests must carry prompt_embeds", flush=True)
        return model.eval() if hasattr(model, "eval") else model


def default_weight_sources(model_dir: str) -> list:
    """transformer/ and vae/ of a diffusers checkpoint (the reference lists the transformer, pipeline_qwen_image.py:246-254,
    and loads the VAE through diffusers, :229-231)."""
    CS = DiffusersPipelineLoader.ComponentSource
    out = [CS(model_or_path=model_dir, subfolder="transformer", prefix="transformer.")]
    if os.path.isdir(os.path.join(model_dir, "vae")):
        out.append(CS(model_or_path=model_dir, subfolder="vae", prefix="vae."))
    return out


def apply_scheduler_config(schedule, cfg: dict) -> None:
    """scheduler_config.json -> FlowMatchConfig fields (the keys the reference reads at pipeline_qwen_image.py:494-500)."""
    c = schedule.config
    for k in ("num_train_timesteps", "base_image_seq_len", "max_image_seq_len", "base_shift", "max_shift", "shift_terminal"):
        if k in cfg and cfg[k] is not None:
            setattr(c, k, type(getattr(c, k) if getattr(c, k) is not None else cfg[k])(cfg[k]))
    if cfg.get("use_dynamic_shifting") is False or cfg.get("time_shift_type", "exponential") != "exponential":
        raise NotImplementedError("only the dynamic exponential time shift of Qwen-Image's scheduler is built")
