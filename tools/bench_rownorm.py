#!/usr/bin/env python3
"""Dev tool (GPU): AdaLN-modulate (rownorm_kernel) and the fp8 row quantiser at the bench step-batch: GB/s of algorithmic traffic."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.devlib  # noqa: E402,F401
import torch  # noqa: E402

from tools.bench_kernels import timeit  # noqa: E402
from vllm_omni_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rows, D, items = 40960, 3072, 10
x = torch.randn(rows, D, device=dev, generator=g).to(torch.bfloat16)
mod = (torch.randn(items, 6 * D, device=dev, generator=g) * 0.3).to(torch.bfloat16)
item = torch.arange(rows, device=dev, dtype=torch.int32) // 4096
yout = torch.empty_like(x)
for blocked in (False, True):
    t = timeit(lambda: ops.adaln_modulate(x, mod[:, D:], mod, mod_item_stride=6 * D, row_item_map=item, out_k32_blocked=blocked, out=yout), iters=50)
    print(f"adaln_modulate {rows}x{D} blocked_out={blocked}: {t * 1e6:7.1f} us = {2 * rows * D * 2 / t / 1e12:.2f} TB/s", flush=True)
for K in (3072, 12288):
    xx = torch.randn(rows, K, device=dev, generator=g).to(torch.bfloat16)
    t = timeit(lambda: ops.quantize_fp8_rows(xx), iters=30)
    print(f"quantize_fp8_rows {rows}x{K}: {t * 1e6:7.1f} us = {3 * rows * K / t / 1e12:.2f} TB/s", flush=True)
