#!/usr/bin/env python3
"""Dev tool: attention kernel only, at the bench shapes (H=24, S=4096+64, B=2 and 6).  Env knobs are read once per
process, so variants are compared by running this script under different OMNI_ATTN_* settings."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
H, S = 24, 4096 + 64
tag = f"PIPE={os.environ.get('OMNI_ATTN_PIPE', '1')} NQ={os.environ.get('OMNI_ATTN_NQ', '1')}"
for B in (2, 6):
    q, k, v = ((torch.randn(B * S, H * 128, device=dev, generator=g)).to(torch.bfloat16) for _ in range(3))
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
    out = ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128))
    q4, k4, v4 = (x.view(B, S, H, 128).permute(0, 2, 1, 3).float() for x in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(q4[:1, :4], k4[:1, :4], v4[:1, :4])
    got = out.view(B, S, H, 128).permute(0, 2, 1, 3)[:1, :4].float()
    err = (got - ref).norm() / ref.norm()
    t = timeit(lambda: ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128)), iters=10)
    fl = 4.0 * B * H * S * S * 128
    print(f"[{tag}] attention B={B}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s  rel_l2 vs fp32 SDPA {err:.2e}", flush=True)
