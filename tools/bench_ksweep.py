#!/usr/bin/env python3
"""Dev tool: GEMM time vs K at fixed (M, N) and epilogue -> per-k-step cost c and per-tile fixed cost X
(prologue + epilogue), from the linear fit  t_tile = c * (K/64) + X.   Tile counts are whole rounds of 256 CUs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
BF16 = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def rn(*shape, s=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * s).to(BF16)


def run(M, N, Ks, epi, name, blocked=False):
    pts = []
    name = name + ("/blk" if blocked else "")
    for K in Ks:
        x, w, b = rn(M, K), rn(N, K, s=0.02), rn(N)
        if blocked:
            w = ops.w_to_k32_blocked(w)
        o = torch.empty(M, N, dtype=BF16, device=dev)
        grp = ops.GemmGroupArgs(x, w, b, o)
        if epi == ops.EPI_BIAS_GATE_RES:
            res, gate = rn(M, N), rn(1, N)
            grp = ops.GemmGroupArgs(x, w, b, o, res=res, gate=gate, gate_item_stride=N, rows_per_item=M)
        t = timeit(lambda: ops.gemm([grp], epi, w_k32_blocked=blocked), iters=20)
        tiles = (M // 256) * (N // 256)
        rounds = -(-tiles // 256)
        us_tile = t * 1e6 / rounds
        pts.append((K // 64, us_tile))
        print(f"{name:10s} M={M} N={N} K={K:6d}: {t*1e6:9.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s  rounds={rounds} us/tile={us_tile:8.2f}", flush=True)
        del x, w, o
    (k0, t0), (k1, t1) = pts[0], pts[-1]
    c = (t1 - t0) / (k1 - k0)
    print(f"  -> fit: c = {c*1e3:.1f} ns per K64-step, X = {t0 - c*k0:.2f} us per tile", flush=True)


if __name__ == "__main__":
    if "--blocked" in sys.argv:      # same-box A/B of the K32-blocked weight layout
        Ks = (1024, 3072, 12288)
        for blk in (False, True, False, True):
            run(4096, 12288, Ks, ops.EPI_BIAS_GELU_TANH, "gelu", blk)
            run(16384, 3072, Ks, ops.EPI_BIAS_GATE_RES, "gate_res", blk)
        sys.exit(0)
    Ks = (512, 1024, 2048, 3072, 6144, 12288)
    run(4096, 12288, Ks, ops.EPI_BIAS, "bias")
    run(4096, 12288, Ks, ops.EPI_BIAS_GELU_TANH, "gelu")
    run(16384, 3072, Ks, ops.EPI_BIAS, "bias")
    run(16384, 3072, Ks, ops.EPI_BIAS_GATE_RES, "gate_res")
    run(8192, 8192, (1024, 8192), ops.EPI_BIAS, "sq")
