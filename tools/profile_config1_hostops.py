#!/usr/bin/env python3
"""dev: which torch-level ops launch the small kernels of one BASELINE config-1 image (256x256, 4 steps, true-CFG, batch 1)?
torch.profiler over ONE warm image: per op name the number of calls, the device time and the Python call site."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd.diffusion.data import OmniDiffusionConfig  # noqa: E402
from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline  # noqa: E402
from vllm_omni_amd.diffusion.request import OmniDiffusionRequest  # noqa: E402

dev = torch.device("cuda:0")
pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(model="x", use_hip_graph=False), device=dev)
pipe.transformer.init_random_(seed=1234)
pipe.vae.init_random_(seed=4321)
g = torch.Generator().manual_seed(3)
req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=4, true_cfg_scale=4.0,
                           latents=torch.randn(1, 256, 64, generator=g).to(dev, torch.bfloat16),
                           prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16),
                           negative_prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16), output_type="latent")
f = lambda: pipe.decode_latents(pipe.generate([req], output_type="latent")[0].output, 256, 256)  # noqa: E731
f(); f()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    f()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60, max_src_column_width=110))
