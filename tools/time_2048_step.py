#!/usr/bin/env python3
"""Dev tool (GPU): one 60-layer DiT forward over a CFG pair of 2048x2048 images (2 x (16384 + 64) rows, BASELINE config 5's geometry),
ms per forward.   [OMNI_DEV_LIB=...] python tools/time_2048_step.py [layers] [n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.devlib  # noqa: E402,F401
from vllm_omni_amd.diffusion.batch import build_ragged_batch  # noqa: E402
from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
BF = torch.bfloat16
m = QwenImageTransformer2DModel(num_layers=layers, device=dev).init_random_(seed=1234)
g = torch.Generator(device=dev).manual_seed(0)
S, T, grid = 16384, 64, (1, 128, 128)
lat = torch.randn(2 * S, 64, device=dev, generator=g).to(BF)
txt = torch.randn(2 * T, 3584, device=dev, generator=g).to(BF)
sig = torch.full((1,), 0.6015625, device=dev)
prep = m.prepare_batch(build_ragged_batch([T, T], grid, temb_rows=[0, 0]))
out = torch.empty(2 * S, 64, dtype=BF, device=dev)
for _ in range(2):
    m.forward_ragged(prep, lat, txt, sig, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    m.forward_ragged(prep, lat, txt, sig, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
dig = int(out.view(torch.int16).to(torch.int64).sum())
print(f"{layers} layers, CFG pair at 2048^2: {dt:.1f} ms per forward (= per denoise step); finite {bool(torch.isfinite(out.float()).all())}; sum of output bits {dig}")
