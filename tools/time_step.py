#!/usr/bin/env python3
"""Dev tool (GPU): the 60-layer DiT forward of ONE true-CFG request (a ragged pair of items) at a given resolution — what a
lightly loaded server runs per denoising step — ms per forward; under rocprofv3 --kernel-trace it is the per-kernel breakdown
of that step.   [OMNI_DEV_LIB=...] python tools/time_step.py <px> [layers] [n] [requests]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.devlib  # noqa: E402,F401
from vllm_omni_amd.diffusion.batch import build_ragged_batch  # noqa: E402
from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel  # noqa: E402

px = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
R = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
BF = torch.bfloat16
m = QwenImageTransformer2DModel(num_layers=layers, device=dev).init_random_(seed=1234)
g = torch.Generator(device=dev).manual_seed(0)
hw = px // 16
S, T, grid = hw * hw, 64, (1, hw, hw)
B = 2 * R
lat = torch.randn(B * S, 64, device=dev, generator=g).to(BF)
txt = torch.randn(B * T, 3584, device=dev, generator=g).to(BF)
sig = torch.full((R,), 0.6015625, device=dev)
prep = m.prepare_batch(build_ragged_batch([T] * B, grid, temb_rows=[i // 2 for i in range(B)]))
out = torch.empty(B * S, 64, dtype=BF, device=dev)
for _ in range(2):
    m.forward_ragged(prep, lat, txt, sig, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    m.forward_ragged(prep, lat, txt, sig, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
# F_fwd per item = L*S*24*D^2 + L*4*S^2*D (+ small terms)
D, Sj = 3072, S + T
flop = B * layers * (Sj * 24 * D * D + 4 * Sj * Sj * D)
dig = int(out.view(torch.int16).to(torch.int64).sum())
print(f"{layers} layers, {R} true-CFG request(s) at {px}^2 ({B} items x {Sj} rows): {dt:.2f} ms per forward (= per denoise step), "
      f"{flop / dt / 1e9:.0f} TF/s = {flop / dt / 1e9 / 2500:.3f} of the bf16 MFMA peak; finite {bool(torch.isfinite(out.float()).all())}; "
      f"sum of output bits {dig}")
