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
for name, N, K, epi in (("mlp_up_gelu", 4 * D, D, ops.EPI_BIAS_GELU_TANH), ("qkv_bias", 3 * D, D, ops.EPI_BIAS)):
    xi, xt, wi, wt, b = rn(Mi, K), rn(Mt, K), rn(N, K, s=0.02), rn(N, K, s=0.02), rn(N)
    xib, xtb, wib, wtb = (ops.w_to_k32_blocked(t) for t in (xi, xt, wi, wt))
    oi, ot = torch.empty(Mi, N, dtype=BF16, device=dev), torch.empty(Mt, N, dtype=BF16, device=dev)
    fl = 2.0 * (Mi + Mt) * N * K
    for rep in range(2):
        for ablk, wblk, oblk in ((0, 0, 0), (1, 1, 0), (1, 1, 1), (0, 0, 1)):
            fn = lambda: ops.gemm([ops.GemmGroupArgs(xib if ablk else xi, wib if wblk else wi, b, oi, a_k32_blocked=bool(ablk), out_k32_blocked=bool(oblk)),  # noqa: E731
                                   ops.GemmGroupArgs(xtb if ablk else xt, wtb if wblk else wt, b, ot, a_k32_blocked=bool(ablk), out_k32_blocked=bool(oblk))],
                                  epi, w_k32_blocked=bool(wblk))
            t = timeit(fn, iters=10)
            print(f"{name}: A_blk={ablk} W_blk={wblk} out_blk={oblk}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF/s", flush=True)
