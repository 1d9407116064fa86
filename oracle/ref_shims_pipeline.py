"""TEST INFRASTRUCTURE — not product code.  Only oracle/gen_golden.py (and the `not gpu` cross-check test that is
skipped when /root/reference is absent) use this.

Shim-import of two more UNMODIFIED reference files, on top of oracle/ref_shims.py:

  * vllm_omni/diffusion/models/qwen_image/autoencoder_kl_qwenimage.py — the vendored Qwen-Image VAE
    (`AutoencoderKLQwenImage.decode`, :839-887; decoder :549-664).  Its diffusers imports are mixins / containers
    with ZERO arithmetic; they are stubbed below (ConfigMixin, register_to_config, ModelMixin, AutoencoderMixin,
    FromOriginalModelMixin, DecoderOutput, AutoencoderKLOutput, DiagonalGaussianDistribution, apply_forward_hook,
    logging) plus `get_activation("silu") -> nn.SiLU()`.
  * vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image.py — for `calculate_shift` (:63-73), `_pack_latents`
    / `_unpack_latents` (:436-457), `prepare_timesteps` (:492-508) and the `diffuse` loop (:530-586) with the
    true-CFG combine (:580-583).  Those are called UNBOUND on a bare object that carries a reference DiT and a
    scheduler.  The one piece of third-party ARITHMETIC is the scheduler:
        diffusers >= 0.36.0 `FlowMatchEulerDiscreteScheduler` (pin: reference pyproject.toml:35)
    which is restated here (`set_timesteps(sigmas=, mu=)` with use_dynamic_shifting / exponential time shift /
    shift_terminal stretch, `step` = fp32 Euler update) — PARITY UNPINNED for that class and for the
    scheduler_config.json values (they live in the HF checkpoint, not in the reference tree).

/root/reference does not exist on the GPU box: never import this from `gpu` tests, bench.py or smoke().
"""
from __future__ import annotations

import importlib
import inspect
import math
import os
import types

import numpy as np
import torch
import torch.nn as nn

import ref_shims
from ref_shims import _mod, _pkg


# ----------------------------------------------------------------------------- diffusers container stubs (no arithmetic)
class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _register_to_config(init):
    sig = inspect.signature(init)

    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_internal_config", _Config(cfg))
        init(self, *args, **kwargs)

    return wrapped


class _ConfigMixin:
    @property
    def config(self):
        return self._internal_config


class _ModelMixin(nn.Module):
    pass


class _Empty:
    pass


class _DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class _AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class _DiagonalGaussianDistribution:
    """mean / logvar split only (the decode path never samples)."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)

    def mode(self):
        return self.mean


def _get_activation(name: str):
    assert name == "silu", name
    return nn.SiLU()


class _Logging:
    @staticmethod
    def get_logger(name):
        import logging

        return logging.getLogger(name)


# ----------------------------------------------------------------------------- third-party arithmetic: the scheduler
class FlowMatchEulerDiscreteSchedulerStub:
    """Restatement of diffusers >= 0.36 FlowMatchEulerDiscreteScheduler for the way the reference pipeline drives it
    (pipeline_qwen_image.py:130,492-508,545,585): `set_timesteps(sigmas=np.ndarray, mu=float)`, `timesteps`,
    `set_begin_index(0)`, `step(model_output, t, sample, return_dict=False)`.  PARITY UNPINNED (see module docstring)."""

    order = 1

    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=0.9,
                 base_image_seq_len=256, max_image_seq_len=8192, shift_terminal=0.02, time_shift_type="exponential"):
        self.config = dict(num_train_timesteps=num_train_timesteps, shift=shift,
                           use_dynamic_shifting=use_dynamic_shifting, base_shift=base_shift, max_shift=max_shift,
                           base_image_seq_len=base_image_seq_len, max_image_seq_len=max_image_seq_len,
                           shift_terminal=shift_terminal, time_shift_type=time_shift_type)
        self._step_index = None
        self._begin_index = None

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, timesteps=None):
        c = self.config
        assert sigmas is not None and mu is not None and c["use_dynamic_shifting"]
        sigmas = np.array(sigmas).astype(np.float32)
        assert c["time_shift_type"] == "exponential"
        sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1) ** 1.0)            # time_shift(mu, 1.0, sigmas)
        if c["shift_terminal"]:
            one_minus_z = 1 - sigmas
            scale_factor = one_minus_z[-1] / (1 - c["shift_terminal"])
            sigmas = 1 - (one_minus_z / scale_factor)
        sigmas = torch.from_numpy(np.asarray(sigmas)).to(dtype=torch.float32, device=device)
        self.timesteps = sigmas * c["num_train_timesteps"]
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self.num_inference_steps = len(self.timesteps)
        self._step_index = None

    def step(self, model_output, timestep, sample, return_dict=True, **_kw):
        if self._step_index is None:
            self._step_index = self._begin_index or 0
        sample = sample.to(torch.float32)
        sigma, sigma_next = self.sigmas[self._step_index], self.sigmas[self._step_index + 1]
        prev = sample + (sigma_next - sigma) * model_output
        self._step_index += 1
        prev = prev.to(model_output.dtype)
        return (prev,)


_installed = False


def install() -> None:
    global _installed
    if _installed:
        return
    ref_shims.install()
    r = os.path.join(ref_shims.REFERENCE_ROOT, "vllm_omni")
    # --- diffusers names the vendored VAE imports
    _mod("diffusers.configuration_utils", ConfigMixin=_ConfigMixin, register_to_config=_register_to_config)
    _mod("diffusers.loaders", FromOriginalModelMixin=_Empty)
    _mod("diffusers.models.activations", get_activation=_get_activation)
    _pkg("diffusers.models.autoencoders")
    _mod("diffusers.models.autoencoders.vae", AutoencoderMixin=type("AutoencoderMixin", (), {}),
         DecoderOutput=_DecoderOutput, DiagonalGaussianDistribution=_DiagonalGaussianDistribution)
    import sys

    sys.modules["diffusers.models.modeling_outputs"].AutoencoderKLOutput = _AutoencoderKLOutput
    _mod("diffusers.models.modeling_utils", ModelMixin=_ModelMixin)
    _pkg("diffusers.utils")
    sys.modules["diffusers.utils"].logging = _Logging
    _mod("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)
    # --- names the reference pipeline module imports at module level
    _mod("diffusers.image_processor", VaeImageProcessor=type("VaeImageProcessor", (), {"__init__": lambda s, **k: None}))
    _mod("diffusers.models.autoencoders.autoencoder_kl_qwenimage", AutoencoderKLQwenImage=object)
    _pkg("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_flow_match_euler_discrete",
         FlowMatchEulerDiscreteScheduler=FlowMatchEulerDiscreteSchedulerStub)
    _mod("diffusers.utils.torch_utils", randn_tensor=lambda shape, generator=None, device=None, dtype=None:
         torch.randn(shape, generator=generator, dtype=dtype).to(device))
    _pkg("vllm.model_executor.models")
    _mod("vllm.model_executor.models.utils", AutoWeightsLoader=object)
    _pkg("vllm_omni.diffusion.model_loader", os.path.join(r, "diffusion", "model_loader"))
    _mod("vllm_omni.diffusion.model_loader.diffusers_loader",
         DiffusersPipelineLoader=type("DiffusersPipelineLoader", (), {"ComponentSource": lambda *a, **k: None}))
    _pkg("vllm_omni.model_executor")
    _pkg("vllm_omni.model_executor.model_loader")
    _mod("vllm_omni.model_executor.model_loader.weight_utils", download_weights_from_hf_specific=lambda *a, **k: None)
    _mod("vllm_omni.diffusion.distributed.utils", get_local_device=lambda: torch.device("cpu"))
    _installed = True


def load_reference_vae_module():
    install()
    return importlib.import_module("vllm_omni.diffusion.models.qwen_image.autoencoder_kl_qwenimage")


def build_reference_vae(dtype=torch.float32):
    """The reference's vendored AutoencoderKLQwenImage with its own defaults (dim_mult passed as a list: the decoder
    concatenates it with a list at :590)."""
    mod = load_reference_vae_module()
    vae = mod.AutoencoderKLQwenImage(dim_mult=[1, 2, 4, 4])
    return vae.to(dtype).eval()


def load_reference_pipeline_module():
    install()
    return importlib.import_module("vllm_omni.diffusion.models.qwen_image.pipeline_qwen_image")


def reference_pipeline_shell(transformer, cfg, scheduler=None):
    """A bare QwenImagePipeline object (no __init__: that would download a checkpoint) carrying what `diffuse` /
    `prepare_timesteps` read: .transformer, .scheduler, ._interrupt, ._attention_kwargs."""
    mod = load_reference_pipeline_module()
    pipe = object.__new__(mod.QwenImagePipeline)
    nn.Module.__init__(pipe)
    pipe.transformer = transformer
    pipe.scheduler = scheduler or FlowMatchEulerDiscreteSchedulerStub()
    pipe._interrupt = False
    pipe._attention_kwargs = None
    pipe._current_timestep = None
    pipe.vae_scale_factor = 8
    pipe._ref_cfg = cfg
    return pipe, mod


def reference_diffuse(pipe, cfg, **kw):
    fc = importlib.import_module("vllm_omni.diffusion.forward_context")
    with torch.no_grad(), fc.set_forward_context(omni_diffusion_config=cfg):
        return type(pipe).diffuse(pipe, **kw)


def load_reference_layered_module():
    """vllm_omni/diffusion/models/qwen_image/pipeline_qwen_image_layered.py, unmodified (its transformers imports are real, its
    diffusers / vllm imports are the stubs installed above)."""
    install()
    return importlib.import_module("vllm_omni.diffusion.models.qwen_image.pipeline_qwen_image_layered")


def reference_layered_shell(transformer, cfg, scheduler=None):
    """A bare QwenImageLayeredPipeline (no __init__: it loads a checkpoint) carrying what `diffuse` reads."""
    mod = load_reference_layered_module()
    pipe = object.__new__(mod.QwenImageLayeredPipeline)
    nn.Module.__init__(pipe)
    pipe.transformer = transformer
    pipe.scheduler = scheduler or FlowMatchEulerDiscreteSchedulerStub()
    pipe._interrupt = False
    pipe._attention_kwargs = None
    pipe._current_timestep = None
    pipe.vae_scale_factor = 8
    return pipe, mod
