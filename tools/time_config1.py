#!/usr/bin/env python3
"""dev: wall time of BASELINE config 1 (256x256, 4 steps, true-CFG, batch 1, 60 layers, + VAE decode) per image."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.devlib  # noqa: E402,F401
from vllm_omni_amd.diffusion.data import OmniDiffusionConfig  # noqa: E402
from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline  # noqa: E402
from vllm_omni_amd.diffusion.request import OmniDiffusionRequest  # noqa: E402

dev = torch.device("cuda:0")
cold = "cold" in sys.argv[1:]          # every image pays its own modulation-table pass (od_config.cache_modulation_tables = False)
sys.argv = [a for a in sys.argv if a not in ("cold", "warm")]
pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(model="x", cache_modulation_tables=not cold), device=dev)
pipe.transformer.init_random_(seed=1234)
pipe.vae.init_random_(seed=4321)
g = torch.Generator().manual_seed(3)
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = (hw // 16) ** 2
req = OmniDiffusionRequest(height=hw, width=hw, num_inference_steps=4, true_cfg_scale=4.0,
                           latents=torch.randn(1, S, 64, generator=g).to(dev, torch.bfloat16),
                           prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16),
                           negative_prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(dev, torch.bfloat16), output_type="latent")
f = lambda: pipe.decode_latents(pipe.generate([req], output_type="latent")[0].output, hw, hw)  # noqa: E731
f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    f()
torch.cuda.synchronize()
print(f"{hw}x{hw}, 4 steps, true-CFG, batch 1: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms/image ({'cold' if cold else 'warm'} schedule table; OMNI_GEMM_SPLITK={os.environ.get('OMNI_GEMM_SPLITK', '1')} SPLITK_INLAUNCH={os.environ.get('OMNI_GEMM_SPLITK_INLAUNCH', '1')} ADALN_PAIR={os.environ.get('OMNI_DIT_ADALN_PAIR', '1')})")
