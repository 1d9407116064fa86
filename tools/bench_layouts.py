#!/usr/bin/env python3
"""Dev tool: MLP-up / QKV / MLP-down at the bench shape (M = 24576 + 384) with each K32-blocked layout switched on/off,
same box, to see what every layout choice costs or buys per GEMM."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
BF16 = torch.bfloat16


def rn(*shape, s=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * s).to(BF16)


D, Mi, Mt = 3072, 24576, 384
SHAPES = (("mlp_up_gelu", 4 * D, D, ops.EPI_BIAS_GELU_TANH), ("qkv_bias", 3 * D, D, ops.EPI_BIAS))
if "--cross" in sys.argv:     # which of N and the epilogue makes an MLP-up round slower than a QKV round?
    SHAPES = (("N12288_bias", 4 * D, D, ops.EPI_BIAS), ("N9216_gelu", 3 * D, D, ops.EPI_BIAS_GELU_TANH),
              ("N12288_gelu", 4 * D, D, ops.EPI_BIAS_GELU_TANH), ("N9216_bias", 3 * D, D, ops.EPI_BIAS),
              ("N6144_bias", 2 * D, D, ops.EPI_BIAS), ("N15360_bias", 5 * D, D, ops.EPI_BIAS))
for name, N, K, epi in SHAPES:
    xi, xt, wi, wt, b = rn(Mi, K), rn(Mt, K), rn(N, K, s=0.02), rn(N, K, s=0.02), rn(N)
    xib, xtb, wib, wtb = (ops.w_to_k32_blocked(t) for t in (xi, xt, wi, wt))
    oi, ot = torch.empty(Mi, N, dtype=BF16, device=dev), torch.empty(Mt, N, dtype=BF16, device=dev)
    fl = 2.0 * (Mi + Mt) * N * K
    for rep in range(2):
        for ablk, wblk, oblk in (((1, 1, 1),) if "--cross" in sys.argv else ((0, 0, 0), (1, 1, 0), (1, 1, 1), (0, 0, 1))):
            fn = lambda: ops.gemm([ops.GemmGroupArgs(xib if ablk else xi, wib if wblk else wi, b, oi, a_k32_blocked=bool(ablk), out_k32_blocked=bool(oblk)),  # noqa: E731
                                   ops.GemmGroupArgs(xtb if ablk else xt, wtb if wblk else wt, b, ot, a_k32_blocked=bool(ablk), out_k32_blocked=bool(oblk))],
                                  epi, w_k32_blocked=bool(wblk))
            t = timeit(fn, iters=10)
            tiles = ((Mi + 255) // 256 + (Mt + 255) // 256) * (N // 256)
            print(f"{name}: A_blk={ablk} W_blk={wblk} out_blk={oblk}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s  "
                  f"({tiles} tiles = {tiles/256:.2f} rounds, {t*1e6/(-(-tiles//256)):.1f} us per round)", flush=True)
