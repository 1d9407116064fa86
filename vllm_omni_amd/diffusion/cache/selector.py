"""Cache backend selection by name — mirror of vllm_omni/diffusion/cache/selector.py:9 (`cache_backend` of
OmniDiffusionConfig / env DIFFUSION_CACHE_BACKEND: "none", "tea_cache"; "cache_dit" is a third-party library adapter that is
not installable here)."""
from __future__ import annotations

from .base import CacheBackend


def get_cache_backend(cache_backend: str | None, cache_config=None) -> CacheBackend | None:
    name, config = cache_backend, cache_config            # (the reference's parameter names: keyword callers keep working)
    if name in (None, "", "none"):
        return None
    if name in ("tea_cache", "teacache"):
        from .teacache.backend import TeaCacheBackend

        return TeaCacheBackend(config)
    if name == "cache_dit":
        raise NotImplementedError("cache-dit is a third-party library (cache-dit==1.1.8) and is not part of this build")
    raise ValueError(f"unknown cache backend {name!r}; supported: none, tea_cache")
