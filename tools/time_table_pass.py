#!/usr/bin/env python3
"""dev: the modulation-table pass (omni_dit_modulation_table: 2 x 60 weight-streaming GEMVs over 13.6 GB) for M conditioning rows."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.devlib  # noqa: E402,F401
from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel  # noqa: E402

dev = torch.device("cuda:0")
m = QwenImageTransformer2DModel(device=dev)
m.init_random_(seed=1234)
gb = 60 * 2 * 18432 * 3072 * 2 / 1e9
for M in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (1, 2, 4, 8):
    sig = torch.linspace(0.9, 0.1, M, device=dev)
    m.modulation_table(sig)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.modulation_table(sig)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"table pass, M = {M} rows: {dt * 1e3:.2f} ms = {dt / 120 * 1e6:.1f} us per 113 MB matrix, {gb / dt / 1e3:.2f} TB/s")
