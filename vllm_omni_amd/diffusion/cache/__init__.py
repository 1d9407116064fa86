"""Step-skipping cache accelerators for the DiT (reference vllm_omni/diffusion/cache/)."""
from .base import CacheBackend
from .selector import get_cache_backend

__all__ = ["CacheBackend", "get_cache_backend"]
