"""Forward-hook registry for diffusion modules — the contract of vllm_omni/diffusion/hooks.py:10-102: a `ModelHook` takes over
a module's forward (`new_forward(module, *args, **kwargs)`), hooks are kept per module in a `HookRegistry` that swaps
`module.forward` for a dispatcher once, and `StateManager` keeps one state object per named context (TeaCache: one per CFG
branch)."""
from __future__ import annotations

from typing import Any, Callable

import torch.nn as nn


class BaseState:
    def reset(self) -> None:
        pass


class StateManager:
    def __init__(self, state_cls: Callable[[], BaseState]):
        self._make, self._states, self._context = state_cls, {}, "default"

    def set_context(self, name: str) -> None:
        self._context = name or "default"

    def get_state(self) -> BaseState:
        if self._context not in self._states:
            self._states[self._context] = self._make()
        return self._states[self._context]

    def reset(self) -> None:
        self._states.clear()


class ModelHook:
    def initialize_hook(self, module: nn.Module) -> nn.Module:
        return module

    def new_forward(self, module: nn.Module, *args: Any, **kwargs: Any):
        raise NotImplementedError

    def reset_state(self, module: nn.Module) -> nn.Module:
        return module


class HookRegistry:
    def __init__(self, module: nn.Module):
        self.module, self._hooks = module, {}

    @classmethod
    def get_or_create(cls, module: nn.Module) -> "HookRegistry":
        reg = getattr(module, "_hook_registry", None)
        if reg is None:
            reg = cls(module)
            module._hook_registry = reg
            if not hasattr(module, "_original_forward"):
                module._original_forward = module.forward
                module.forward = reg.dispatch            # instance attribute shadows the class method: nn.Module calls it
        return reg

    def register_hook(self, name: str, hook: ModelHook) -> None:
        hook.initialize_hook(self.module)
        self._hooks[name] = hook

    def remove_hook(self, name: str) -> None:
        self._hooks.pop(name, None)

    def get_hook(self, name: str) -> ModelHook | None:
        return self._hooks.get(name)

    def dispatch(self, *args: Any, **kwargs: Any):
        if not self._hooks:
            return self.module._original_forward(*args, **kwargs)
        return self._hooks[sorted(self._hooks)[0]].new_forward(self.module, *args, **kwargs)

    def reset_hook(self, name: str) -> None:
        hook = self._hooks.get(name)
        if hook is not None:
            hook.reset_state(self.module)
