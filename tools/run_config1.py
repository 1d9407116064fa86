#!/usr/bin/env python3
"""dev: BASELINE config 1 (256x256, 4 steps, true-CFG, batch 1, 60 layers) a few times, for rocprofv3 --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd.diffusion.data import OmniDiffusionConfig  # noqa: E402
from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline  # noqa: E402
from vllm_omni_amd.diffusion.request import OmniDiffusionRequest  # noqa: E402

dev = torch.device("cuda:0")
pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(model="x"), device=dev)
pipe.transformer.init_random_(seed=1234)
pipe.vae.init_random_(seed=4321)
g = torch.Generator().manual_seed(3)
req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=4, true_cfg_scale=4.0,
                           latents=torch.randn(1, 256, 64, generator=g).to(dev, torch.bfloat16),
                           prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16),
                           negative_prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16), output_type="latent")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    pipe.decode_latents(pipe.generate([req], output_type="latent")[0].output, 256, 256)
torch.cuda.synchronize()
print("done")
