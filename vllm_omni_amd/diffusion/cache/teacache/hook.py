"""Hook-based TeaCache (host-driven) — the behaviour of vllm_omni/diffusion/cache/teacache/hook.py:26-257 on this
transformer's module surface: the hook replaces `transformer.forward`, asks the extractor for a CacheContext, decides from
the relative L1 distance of consecutive modulated inputs (polynomial rescale, accumulate, threshold; one state per CFG
branch) and either reuses the cached residual or runs the block stack.

This is the COMPATIBILITY path (it keeps the reference's host read-back per forward and runs one native call per block); the
production path is the device-side TeaCache inside `omni_dit_forward` (native.py).  Both implement the same decision rule."""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from ...hooks import BaseState, HookRegistry, ModelHook, StateManager
from .config import TeaCacheConfig
from .extractors import get_extractor


class TeaCacheState(BaseState):
    def __init__(self):
        self.reset()

    def reset(self) -> None:
        self.cnt = 0
        self.accumulated_rel_l1_distance = 0.0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.previous_residual_encoder = None


class TeaCacheHook(ModelHook):
    _HOOK_NAME = "teacache"

    def __init__(self, config: TeaCacheConfig):
        self.config = config
        self.rescale_func = np.poly1d(config.coefficients)
        self.state_manager = StateManager(TeaCacheState)
        self.extractor_fn = None
        self._forward_cnt = 0
        self.decisions: list[bool] = []          # True = computed (kept for tests / statistics)
        self.rescaled_history: list[float] = []  # |poly(rel)| of every decided forward

    def initialize_hook(self, module):
        self.extractor_fn = get_extractor(self.config.transformer_type)
        self.state_manager.set_context("teacache")
        return module

    def new_forward(self, module, *args: Any, **kwargs: Any):
        ctx = self.extractor_fn(module, *args, **kwargs)
        branch = "negative" if (getattr(module, "do_true_cfg", False) and self._forward_cnt % 2 == 1) else "positive"
        self.state_manager.set_context(f"teacache_{branch}")
        state = self.state_manager.get_state()
        compute = self._should_compute_full_transformer(state, ctx.modulated_input)
        if not compute and state.previous_residual is not None:
            ctx.hidden_states = ctx.hidden_states + state.previous_residual
            output = ctx.hidden_states
        else:
            compute = True
            ori = ctx.hidden_states.clone()
            outputs = ctx.run_transformer_blocks()
            ctx.hidden_states = outputs[0]
            state.previous_residual = (ctx.hidden_states - ori).detach()
            output = ctx.hidden_states
        self.decisions.append(compute)
        state.previous_modulated_input = ctx.modulated_input.detach()
        state.cnt += 1
        self._forward_cnt += 1
        return ctx.postprocess(output)

    def _should_compute_full_transformer(self, state: TeaCacheState, modulated_inp: torch.Tensor) -> bool:
        if state.cnt == 0:
            state.accumulated_rel_l1_distance = 0.0
            return True
        if state.previous_modulated_input is None:
            return True
        prev = state.previous_modulated_input
        rel = ((modulated_inp - prev).abs().mean() / (prev.abs().mean() + 1e-8)).cpu().item()
        self.rescaled_history.append(abs(float(self.rescale_func(rel))))
        state.accumulated_rel_l1_distance += self.rescaled_history[-1]
        if state.accumulated_rel_l1_distance < self.config.rel_l1_thresh:
            return False
        state.accumulated_rel_l1_distance = 0.0
        return True

    def reset_state(self, module):
        self.state_manager.reset()
        self._forward_cnt = 0
        self.decisions, self.rescaled_history = [], []
        return module


def apply_teacache_hook(module: torch.nn.Module, config: TeaCacheConfig) -> TeaCacheHook:
    hook = TeaCacheHook(config)
    HookRegistry.get_or_create(module).register_hook(TeaCacheHook._HOOK_NAME, hook)
    return hook
