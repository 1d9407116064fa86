"""Config / output dataclasses — the subset of vllm_omni/diffusion/data.py the DiT path touches
(DiffusionParallelConfig :26-91, TransformerConfig :95-117, OmniDiffusionConfig :237-455, DiffusionOutput :508-520).
Field names and defaults follow the reference so configs written for it construct here unchanged."""
from __future__ import annotations

from collections.abc import Callable
from contextlib import contextmanager
from dataclasses import dataclass, field
from typing import Any

import torch


_PARALLEL_AXES = ("pipeline_parallel_size", "data_parallel_size", "tensor_parallel_size", "ulysses_degree",
                  "ring_degree", "cfg_parallel_size")


@dataclass
class DiffusionParallelConfig:
    """Degrees of every parallel axis the reference's config knows (same field names as
    vllm_omni/diffusion/data.py:26-91).  This build shards by REQUEST (`data_parallel_size`) and,
    for single-image latency, by SEQUENCE with Ulysses (`ulysses_degree`, §8f N2); every other axis must stay 1."""
    pipeline_parallel_size: int = 1
    data_parallel_size: int = 1
    tensor_parallel_size: int = 1
    sequence_parallel_size: int | None = None      # = ulysses_degree * ring_degree
    ulysses_degree: int = 1
    ring_degree: int = 1
    cfg_parallel_size: int = 1

    def __post_init__(self) -> None:
        sp = self.ulysses_degree * self.ring_degree
        if self.sequence_parallel_size is None:
            self.sequence_parallel_size = sp
        bad = [ax for ax in _PARALLEL_AXES if int(getattr(self, ax)) <= 0]
        if bad:
            raise AssertionError(f"parallel degrees must be positive: {bad}")
        if self.sequence_parallel_size != sp:
            raise AssertionError(f"sequence_parallel_size {self.sequence_parallel_size} != ulysses {self.ulysses_degree}"
                                 f" * ring {self.ring_degree}")
        unsupported = [ax for ax in _PARALLEL_AXES if ax not in ("data_parallel_size", "ulysses_degree") and getattr(self, ax) != 1]
        if unsupported:
            raise NotImplementedError(f"{unsupported} > 1 is not built; the MI355X path is data-parallel over requests, "
                                      "optionally Ulysses sequence-parallel inside a request")
        self.world_size = self.data_parallel_size * self.ulysses_degree

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "DiffusionParallelConfig":
        if not isinstance(data, dict):
            raise TypeError(f"Expected parallel config dict, got {type(data)!r}")
        return cls(**data)


class TransformerConfig(dict):
    """`transformer/config.json` as a mapping with attribute access (role of vllm_omni/diffusion/data.py:94-117; the DiT
    reads `num_layers` only, qwen_image_transformer.py:652-653).  A plain dict subclass: `TransformerConfig(num_layers=2)`,
    `TransformerConfig.from_dict({...})`, `.get(k, default)`, `.num_layers`, `.to_dict()`, `.params`."""

    @classmethod
    def from_dict(cls, data: dict[str, Any]) -> "TransformerConfig":
        if not isinstance(data, dict):
            raise TypeError(f"Expected transformer config dict, got {type(data)!r}")
        return cls(data)

    def to_dict(self) -> dict[str, Any]:
        return dict(self)

    @property
    def params(self) -> "TransformerConfig":
        return self

    def __getattr__(self, item: str) -> Any:
        try:
            return self[item]
        except KeyError as exc:
            raise AttributeError(item) from exc


@dataclass
class OmniDiffusionConfig:
    model: str | None = None
    model_class_name: str | None = "QwenImagePipeline"
    dtype: torch.dtype = torch.bfloat16
    tf_model_config: TransformerConfig = field(default_factory=TransformerConfig)
    parallel_config: DiffusionParallelConfig = field(default_factory=DiffusionParallelConfig)
    cache_backend: str = "none"
    cache_config: dict[str, Any] = field(default_factory=dict)
    num_gpus: int | None = None
    master_port: int | None = None
    output_type: str = "pil"
    max_step_batch: int = 5          # NEW (not in the reference): requests whose steps share one DiT forward (5 x 2 CFG
                                     # items x 4160 rows fills the 256 CUs' tile rounds best at 1024^2: DESIGN.md 7)
    dist_timeout: int | None = None
    use_hip_graph: bool | None = None   # NEW: capture one denoise step as a hipGraph (None = automatic by size)
    devices: list[int] | None = None # device index of every rank (None: rank r -> cuda:r).  The reference maps a stage's
                                     # `runtime.devices` list the same way (entrypoints/stage_utils.py:14-176); several ranks
                                     # MAY share a device (then use dist_backend="gloo": RCCL refuses duplicate GPUs)
    dist_backend: str | None = None  # None: "nccl" (= RCCL) with GPUs, "gloo" without
    load_text_encoder: bool = True   # NEW: False = never build the ~16 GB Qwen2.5-VL prompt encoder on the workers (requests
                                     # then carry prompt_embeds)
    precompute_modulation: bool = True  # NEW: compute every block's modulation vectors for ALL steps of a request in one pass
                                     # over the modulation weights (omni_dit_modulation_table) instead of re-streaming 13.6 GB
                                     # of weights in every forward
    cache_modulation_tables: bool = True  # NEW: keep a schedule's modulation table for later requests with the same sigmas
                                     # (the same resolution and step count): a function of weights + schedule only, never of
                                     # a request's prompt / latents / seed (QwenImageTransformer2DModel.modulation_table_for_schedule)
    max_steps_in_flight: int = 2     # NEW: how many denoising steps a worker's host may enqueue ahead of the device
                                     # (step_batcher.py: bounded run-ahead, so that a newcomer joins within this many steps)
    # ---- the reference's remaining fields (data.py:255-360), so that a config written for it constructs here unchanged.
    # Inert here (nothing on this path depends on them, results are the same either way): executor / server plumbing, HF hub
    # options, CPU-offload and pinning switches (41 GB of weights stay resident in 288 GB of HBM), VAE slicing, torch.compile
    # (there is no tracing compiler: kernels are hand-written), logging.  Fields that would CHANGE results if honoured — LoRA,
    # FSDP / HSDP sharding, the sparse-attention variants, VAE tiling, another transformer class, the Wan-only schedule knobs —
    # are refused when set away from the reference's defaults (`__post_init__`), never silently ignored.
    cache_strategy: str = "none"
    distributed_executor_backend: str = "mp"
    nccl_port: int | None = None
    trust_remote_code: bool = False
    revision: str | None = None
    hsdp_replicate_dim: int = 1
    hsdp_shard_dim: int = -1
    lora_path: str | None = None
    lora_nickname: str = "default"
    lora_target_modules: list[str] | None = None
    dit_cpu_offload: bool = True
    use_fsdp_inference: bool = False
    text_encoder_cpu_offload: bool = True
    image_encoder_cpu_offload: bool = True
    vae_cpu_offload: bool = True
    pin_cpu_memory: bool = True
    vae_use_slicing: bool = False
    vae_use_tiling: bool = False
    mask_strategy_file_path: str | None = None
    skip_time_steps: int = 15
    enable_torch_compile: bool = False
    disable_autocast: bool = False
    VSA_sparsity: float = 0.0
    moba_config_path: str | None = None
    host: str | None = None
    port: int | None = None
    scheduler_port: int = 5555
    enable_stage_verification: bool = True
    prompt_file_path: str | None = None
    model_paths: dict[str, str] = field(default_factory=dict)
    model_loaded: dict[str, bool] = field(default_factory=lambda: {"transformer": True, "vae": True})
    override_transformer_cls_name: str | None = None
    boundary_ratio: float | None = None
    flow_shift: float | None = None
    supports_multimodal_inputs: bool = False
    log_level: str = "info"

    _REFUSED = {"lora_path": None, "use_fsdp_inference": False, "hsdp_replicate_dim": 1, "hsdp_shard_dim": -1,
                "mask_strategy_file_path": None, "VSA_sparsity": 0.0, "moba_config_path": None,
                "override_transformer_cls_name": None, "boundary_ratio": None, "flow_shift": None}

    def __post_init__(self):
        changed = sorted(k for k, dflt in self._REFUSED.items() if getattr(self, k) != dflt)
        if changed:
            raise NotImplementedError(f"{changed}: not built on the MI355X path, and honouring them would change the results — "
                                      "refused instead of ignored")
        if isinstance(self.parallel_config, dict):
            self.parallel_config = DiffusionParallelConfig.from_dict(self.parallel_config)
        if not isinstance(self.tf_model_config, TransformerConfig):
            self.tf_model_config = TransformerConfig.from_dict(self.tf_model_config)
        if self.num_gpus is None:
            self.num_gpus = self.parallel_config.world_size
        if self.cache_backend is None:
            self.cache_backend = "none"
        if self.cache_backend not in ("none", "tea_cache", "teacache"):
            raise NotImplementedError(f"cache backend {self.cache_backend!r}: only 'none' and 'tea_cache' are built "
                                      "(cache-dit is a third-party library)")


_current: OmniDiffusionConfig | None = None


@contextmanager
def set_current_omni_diffusion_config(cfg: OmniDiffusionConfig, check_compile=False, prefix: str | None = None):
    global _current
    old = _current
    _current = cfg
    try:
        yield
    finally:
        _current = old


def get_current_omni_diffusion_config() -> OmniDiffusionConfig:
    return _current if _current is not None else OmniDiffusionConfig()


@dataclass
class DiffusionOutput:
    output: torch.Tensor | None = None
    trajectory_timesteps: list[torch.Tensor] | None = None
    trajectory_latents: torch.Tensor | None = None
    trajectory_decoded: list[torch.Tensor] | None = None
    error: str | None = None
    post_process_func: Callable[..., Any] | None = None
