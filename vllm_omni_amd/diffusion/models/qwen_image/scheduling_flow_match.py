"""Flow-Match Euler schedule used by Qwen-Image (host side).

Restates what the reference gets from diffusers' FlowMatchEulerDiscreteScheduler through
prepare_timesteps / retrieve_timesteps (pipeline_qwen_image.py:492-508) with Qwen-Image's scheduler_config
(use_dynamic_shifting, exponential time shift, shift_terminal 0.02): sigmas = linspace(1, 1/N, N) ->
sigma' = e^mu / (e^mu + (1/sigma - 1)) with mu linear in the image token count -> stretched so the last sigma
equals shift_terminal -> 0 appended.  The per-step update x <- x + (sigma_{i+1} - sigma_i) * v runs on the GPU
fused with the CFG combine (omni_cfg_euler_step)."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class FlowMatchConfig:
    num_train_timesteps: int = 1000
    base_image_seq_len: int = 256
    max_image_seq_len: int = 8192
    base_shift: float = 0.5
    max_shift: float = 0.9
    shift_terminal: float | None = 0.02


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


class FlowMatchEulerSchedule:
    def __init__(self, config: FlowMatchConfig | None = None):
        self.config = config or FlowMatchConfig()
        self.timesteps: torch.Tensor | None = None
        self.sigmas: torch.Tensor | None = None

    def set_timesteps(self, num_inference_steps: int, image_seq_len: int, sigmas=None, mu: float | None = None):
        """`mu` given: used as is (the Layered pipeline passes sqrt(S_cond / 256), pipeline_qwen_image_layered.py:808-816);
        else linear in `image_seq_len` (calculate_shift)."""
        c = self.config
        if num_inference_steps is None or num_inference_steps < 1:
            raise ValueError("num_inference_steps must be >= 1")
        s = np.linspace(1.0, 1.0 / num_inference_steps, num_inference_steps) if sigmas is None else np.asarray(sigmas)
        if mu is None:
            mu = calculate_shift(image_seq_len, c.base_image_seq_len, c.max_image_seq_len, c.base_shift, c.max_shift)
        s = np.array(s).astype(np.float32)                    # diffusers computes the schedule in float32 numpy
        s = math.exp(mu) / (math.exp(mu) + (1 / s - 1) ** 1.0)
        if c.shift_terminal:
            one_minus = 1 - s
            # a single step has sigma = 1: nothing to stretch (the diffusers formula is 0/0 = NaN there)
            if float(one_minus[-1]) > 0.0:
                s = 1 - (one_minus / (one_minus[-1] / (1 - c.shift_terminal)))
        sig = torch.from_numpy(np.asarray(s)).to(torch.float32)
        self.timesteps = sig * c.num_train_timesteps
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        return self.timesteps

    def dt(self) -> torch.Tensor:
        """sigma_{i+1} - sigma_i, fp32 [N]."""
        return self.sigmas[1:] - self.sigmas[:-1]

    @staticmethod
    def model_timestep(t: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
        """What the DiT receives for scheduler timestep t: the reference does
        `t.expand(B).to(latents.dtype)` then `/ 1000` in that dtype (pipeline_qwen_image.py:552,558), and the
        model casts to the activation dtype again (qwen_image_transformer.py:746).  Returned as fp32."""
        return (t.to(dtype) / 1000).to(dtype).to(torch.float32)
