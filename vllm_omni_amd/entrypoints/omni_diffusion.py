"""OmniDiffusion — the stage-engine entry point of vllm_omni/entrypoints/omni_diffusion.py:32-90: build an
OmniDiffusionConfig (from kwargs or given), start a DiffusionEngine, turn `generate(prompt, **sampling)` into
OmniDiffusionRequests (unknown kwargs are dropped, as in `prepare_requests`, :20-29) and return OmniRequestOutput(s).

Reading `model_index.json` / `transformer/config.json` from a checkpoint directory works when `model` is a local path (no
network here); otherwise the defaults of the Qwen-Image architecture are used and the weights are what the pipeline factory
provides (random-init in the tests)."""
from __future__ import annotations

import json
import os
from dataclasses import fields

from ..diffusion.data import OmniDiffusionConfig, TransformerConfig
from ..diffusion.diffusion_engine import DiffusionEngine
from ..diffusion.request import OmniDiffusionRequest


def prepare_requests(prompt, **kwargs) -> OmniDiffusionRequest:
    names = {f.name for f in fields(OmniDiffusionRequest)}
    init = {"prompt": prompt}
    init.update({k: v for k, v in kwargs.items() if k in names})
    return OmniDiffusionRequest(**init)


class OmniDiffusion:
    def __init__(self, od_config: OmniDiffusionConfig | dict | None = None, pipeline_factory=None, **kwargs):
        if od_config is None:
            od_config = OmniDiffusionConfig(**{k: v for k, v in kwargs.items() if k in {f.name for f in fields(OmniDiffusionConfig)}})
        elif isinstance(od_config, dict):
            od_config = OmniDiffusionConfig(**od_config)
        self.od_config = od_config
        model = od_config.model
        if model and os.path.isdir(model):
            idx = os.path.join(model, "model_index.json")
            if os.path.exists(idx):
                od_config.model_class_name = json.load(open(idx)).get("_class_name", od_config.model_class_name)
            tf = os.path.join(model, "transformer", "config.json")
            if os.path.exists(tf):
                od_config.tf_model_config = TransformerConfig.from_dict(json.load(open(tf)))
        self.engine = DiffusionEngine.make_engine(od_config, pipeline_factory=pipeline_factory)

    def generate(self, prompt, **kwargs):
        if isinstance(prompt, str):
            prompts = [prompt]
        elif isinstance(prompt, list):
            prompts = list(prompt)
        else:
            raise ValueError("Prompt must be a string or a list of strings")
        return self.engine.step([prepare_requests(p, **kwargs) for p in prompts])

    def close(self) -> None:
        self.engine.close()

    def __del__(self):  # pragma: no cover - best effort cleanup
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
