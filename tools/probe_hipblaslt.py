#!/usr/bin/env python3
"""dev tool: run torch.mm (hipBLASLt) at the DiT GEMM shapes so that `rocprofv3 --kernel-trace` shows which kernel
configuration (macro tile, MFMA shape, LDS use, waves) the vendor library picks — a known-good on-hardware reference."""
import torch
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
D = 3072
for M in (24576, 8192):
    for name, N, K in (("qkv", 3 * D, D), ("out", D, D), ("up", 4 * D, D), ("down", D, 4 * D)):
        a = torch.randn(M, K, device=dev, generator=g).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
        for _ in range(3):
            y = torch.mm(a, w.t())
        torch.cuda.synchronize()
a = torch.randn(8192, 8192, device=dev, generator=g).bfloat16()
for _ in range(3):
    y = torch.mm(a, a.t())
torch.cuda.synchronize()
