"""TeaCacheBackend — mirror of vllm_omni/diffusion/cache/teacache/backend.py:24-113.

`enable(pipeline)` validates the configuration for the pipeline's transformer class and switches the pipeline's denoise loop
to the device-side TeaCache path (native.py); `refresh` resets the per-generation state.  (The reference installs a forward
hook that re-walks the model in Python; the module-level surface that hook needs exists on this transformer too — see
hook.py — but the production path keeps the whole forward in ONE native call.)"""
from __future__ import annotations

from typing import Any

from ..base import CacheBackend
from .config import TeaCacheConfig


class TeaCacheBackend(CacheBackend):
    def enable(self, pipeline: Any) -> None:
        transformer = pipeline.transformer
        cfg = self.config if isinstance(self.config, dict) else getattr(self.config, "__dict__", {})
        try:
            tc = TeaCacheConfig(transformer_type=transformer.__class__.__name__,
                                rel_l1_thresh=cfg.get("rel_l1_thresh", 0.2), coefficients=cfg.get("coefficients"))
        except Exception as e:
            raise ValueError(f"Invalid TeaCache configuration: {e}. Expected keys: rel_l1_thresh, coefficients (optional).") from e
        transformer.teacache = tc
        self.enabled = True

    def refresh(self, pipeline: Any, num_inference_steps: int, verbose: bool = True) -> None:
        """New generation (reference backend.py:96-113, called by the worker before every request batch,
        worker/gpu_worker.py:132-134): every resident TeaCache device state of the static denoise loop goes back to
        "first forward computes" (counters and accumulated distances to zero: native.TeaCacheDeviceState.reset).  The serving
        path keeps history per SAMPLE (a fresh sample is imported with zeroed counters), so its states are left alone —
        requests that are mid-loop keep theirs."""
        self.num_inference_steps = int(num_inference_steps)
        for st in list(getattr(pipeline, "_step_state", {}).values()):
            tc = st.get("tc") if isinstance(st, dict) else None
            if tc is not None:
                tc.reset()
