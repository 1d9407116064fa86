"""TeaCacheConfig — same fields / validation as vllm_omni/diffusion/cache/teacache/config.py:34-72.  The polynomial that
rescales the relative L1 distance is model specific; the Qwen-Image coefficients are the published ComfyUI-TeaCache tuning
the reference ships (config.py:21-31), highest power first (numpy.poly1d order)."""
from __future__ import annotations

from dataclasses import dataclass

MODEL_COEFFICIENTS = {
    "QwenImageTransformer2DModel": [-4.50000000e02, 2.80000000e02, -4.50000000e01, 3.20000000e00, -2.00000000e-02],
}


@dataclass
class TeaCacheConfig:
    rel_l1_thresh: float = 0.2
    coefficients: list[float] | None = None
    transformer_type: str = "QwenImageTransformer2DModel"

    def __post_init__(self) -> None:
        if self.rel_l1_thresh <= 0:
            raise ValueError(f"rel_l1_thresh must be positive, got {self.rel_l1_thresh}")
        if self.coefficients is None:
            if self.transformer_type not in MODEL_COEFFICIENTS:
                raise KeyError(f"Cannot find coefficients for {self.transformer_type}. Supported: {list(MODEL_COEFFICIENTS)}")
            self.coefficients = list(MODEL_COEFFICIENTS[self.transformer_type])
        if len(self.coefficients) != 5:
            raise ValueError(f"coefficients must contain exactly 5 elements, got {len(self.coefficients)}")
