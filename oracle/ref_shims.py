"""TEST INFRASTRUCTURE — not product code.  Only oracle/gen_golden.py uses this.

Shim-import of the UNMODIFIED reference DiT.

`vllm` and `diffusers` are not installed in the authoring container, so
`import vllm_omni` fails at vllm_omni/config/model.py:6.  This module registers
bare package objects for the `vllm_omni.*` packages (so their `__init__.py`,
which pull in vLLM, are skipped) with `__path__` pointing INTO /root/reference,
and provides stub modules that restate only the third-party *leaf ops* the DiT
file imports:

  vllm v0.12.0   (pin: reference docker/Dockerfile.rocm:5)
    - model_executor.layers.layernorm.RMSNorm          (native path: fp32 var, cast, * weight)
    - model_executor.layers.linear.ReplicatedLinear    -> (F.linear(x, W, b), None)
    - model_executor.layers.linear.QKVParallelLinear   -> fused [q|k|v] rows, weight_loader(shard_id)
    - model_executor.model_loader.weight_utils.default_weight_loader
  diffusers >= 0.36.0 (pin: reference pyproject.toml:35)
    - models.attention.FeedForward("gelu-approximate") = Linear -> GELU(tanh) -> Linear
    - models.embeddings.Timesteps / TimestepEmbedding  (body duplicated in-tree at
      vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image.py:135-184)
    - models.normalization.AdaLayerNormContinuous
    - models.modeling_outputs.Transformer2DModelOutput

Everything else (block control flow, RoPE tables, AdaLayerNorm, RotaryEmbedding,
Attention + SDPA backend, config dataclasses) is the reference's own code, loaded
from /root/reference at run time.  Nothing is copied into this repository.

/root/reference does not exist on the GPU box: this file must never be imported
by tests marked `gpu`, by bench.py or by __graft_entry__.smoke().
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("OMNI_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vllm_omni", "diffusion"))


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name: str, path: str | None = None) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []  # type: ignore[attr-defined]
    sys.modules[name] = m
    return m


# ----------------------------------------------------------------------------- vllm stubs
class _RMSNorm(nn.Module):
    """vllm RMSNorm.forward_native: x.float(); x*rsqrt(mean(x^2)+eps); .to(dtype); * weight."""

    def __init__(self, hidden_size: int, eps: float = 1e-6, **_kw):
        super().__init__()
        self.variance_epsilon = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        dt = x.dtype
        xf = x.float()
        var = xf.pow(2).mean(dim=-1, keepdim=True)
        xf = xf * torch.rsqrt(var + self.variance_epsilon)
        return xf.to(dt) * self.weight


class _ReplicatedLinear(nn.Module):
    def __init__(self, input_size: int, output_size: int, bias: bool = True, **_kw):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(output_size, input_size))
        self.bias = nn.Parameter(torch.zeros(output_size)) if bias else None

    def forward(self, x):
        return F.linear(x, self.weight, self.bias), None


class _QKVParallelLinear(nn.Module):
    """Fused q|k|v projection (rows ordered q, k, v), bias on, tuple return."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = True, disable_tp: bool = False, **_kw):
        super().__init__()
        kvh = total_num_kv_heads or total_num_heads
        self.q_size = total_num_heads * head_size
        self.kv_size = kvh * head_size
        out = self.q_size + 2 * self.kv_size
        self.weight = nn.Parameter(torch.empty(out, hidden_size))
        self.bias = nn.Parameter(torch.zeros(out)) if bias else None
        for p in (self.weight, self.bias):
            if p is not None:
                p.weight_loader = self.weight_loader  # type: ignore[attr-defined]

    def weight_loader(self, param, loaded_weight, shard_id=None):
        off = {"q": 0, "k": self.q_size, "v": self.q_size + self.kv_size}
        size = {"q": self.q_size, "k": self.kv_size, "v": self.kv_size}
        if shard_id is None:
            param.data.copy_(loaded_weight)
        else:
            param.data[off[shard_id]: off[shard_id] + size[shard_id]].copy_(loaded_weight)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias), None


def _default_weight_loader(param, loaded_weight):
    param.data.copy_(loaded_weight)


class _Platform:
    device_type = "cpu"

    @staticmethod
    def is_rocm():
        return False

    @staticmethod
    def is_cuda():
        return False


# ----------------------------------------------------------------------------- diffusers stubs
class _GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class _FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", **_kw):
        super().__init__()
        assert activation_fn == "gelu-approximate"
        inner = int(dim * mult)
        self.net = nn.ModuleList([_GELU(dim, inner, approximate="tanh"), nn.Dropout(dropout),
                                  nn.Linear(inner, dim_out or dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class _Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.shift)
        emb = torch.exp(exponent).to(timesteps.dtype)
        emb = timesteps[:, None].float() * emb[None, :]
        emb = self.scale * emb
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, **_kw):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class _AdaLayerNormContinuous(nn.Module):
    def __init__(self, embedding_dim, conditioning_embedding_dim, elementwise_affine=True, eps=1e-5, bias=True, **_kw):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_embedding_dim, embedding_dim * 2, bias=bias)
        self.norm = nn.LayerNorm(embedding_dim, eps, elementwise_affine, bias)

    def forward(self, x, conditioning_embedding):
        emb = self.linear(self.silu(conditioning_embedding).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class _Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


_installed = False


def install() -> None:
    """Register stubs + bare reference packages in sys.modules (idempotent)."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    import logging

    # vllm
    _pkg("vllm")
    _mod("vllm.logger", init_logger=lambda name: logging.getLogger(name))
    _mod("vllm.platforms", current_platform=_Platform())
    _pkg("vllm.config")
    sys.modules["vllm.config"].VllmConfig = type("VllmConfig", (), {})
    _mod("vllm.config.utils", config=lambda cls: cls)
    _pkg("vllm.distributed")
    _mod("vllm.distributed.parallel_state", _TP=None, get_tensor_model_parallel_world_size=lambda: 1)
    _pkg("vllm.model_executor")
    _pkg("vllm.model_executor.layers")
    _mod("vllm.model_executor.layers.layernorm", RMSNorm=_RMSNorm)
    _mod("vllm.model_executor.layers.linear", ReplicatedLinear=_ReplicatedLinear, QKVParallelLinear=_QKVParallelLinear)
    _pkg("vllm.model_executor.model_loader")
    _mod("vllm.model_executor.model_loader.weight_utils", default_weight_loader=_default_weight_loader)
    # diffusers
    _pkg("diffusers")
    _pkg("diffusers.models")
    _mod("diffusers.models.attention", FeedForward=_FeedForward)
    _mod("diffusers.models.embeddings", Timesteps=_Timesteps, TimestepEmbedding=_TimestepEmbedding)
    _mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=_Transformer2DModelOutput)
    _mod("diffusers.models.normalization", AdaLayerNormContinuous=_AdaLayerNormContinuous)
    # bare reference packages (skip their __init__.py)
    r = os.path.join(REFERENCE_ROOT, "vllm_omni")
    _pkg("vllm_omni", r)
    _pkg("vllm_omni.utils", os.path.join(r, "utils"))
    _pkg("vllm_omni.diffusion", os.path.join(r, "diffusion"))
    _pkg("vllm_omni.diffusion.utils", os.path.join(r, "diffusion", "utils"))
    _pkg("vllm_omni.diffusion.layers", os.path.join(r, "diffusion", "layers"))
    _pkg("vllm_omni.diffusion.cache", os.path.join(r, "diffusion", "cache"))
    _pkg("vllm_omni.diffusion.distributed", os.path.join(r, "diffusion", "distributed"))
    _pkg("vllm_omni.diffusion.attention", os.path.join(r, "diffusion", "attention"))
    _pkg("vllm_omni.diffusion.attention.backends", os.path.join(r, "diffusion", "attention", "backends"))
    _pkg("vllm_omni.diffusion.models", os.path.join(r, "diffusion", "models"))
    _pkg("vllm_omni.diffusion.models.qwen_image", os.path.join(r, "diffusion", "models", "qwen_image"))
    _installed = True


def load_reference_dit():
    """Return the reference module object for qwen_image_transformer.py (unmodified)."""
    install()
    # attention.parallel has a real __init__ that is import-clean once the stubs exist
    return importlib.import_module("vllm_omni.diffusion.models.qwen_image.qwen_image_transformer")


def build_reference_model(num_layers: int, *, num_attention_heads: int = 24, attention_head_dim: int = 128,
                          joint_attention_dim: int = 3584, dtype=torch.float32, **model_kw):
    """Instantiate the reference QwenImageTransformer2DModel on CPU (reference file :609-690)."""
    mod = load_reference_dit()
    data = importlib.import_module("vllm_omni.diffusion.data")
    cfg = data.OmniDiffusionConfig(model="x", dtype=dtype,
                                   tf_model_config=data.TransformerConfig.from_dict({"num_layers": num_layers}),
                                   num_gpus=1)
    with data.set_current_omni_diffusion_config(cfg):
        model = mod.QwenImageTransformer2DModel(od_config=cfg, num_attention_heads=num_attention_heads,
                                                attention_head_dim=attention_head_dim,
                                                joint_attention_dim=joint_attention_dim, **model_kw)
    return model.to(dtype).eval(), cfg


def reference_forward(model, cfg, **kw):
    fc = importlib.import_module("vllm_omni.diffusion.forward_context")
    with torch.no_grad(), fc.set_forward_context(omni_diffusion_config=cfg):
        return model(**kw, return_dict=False)[0]
