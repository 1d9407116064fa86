#!/usr/bin/env python3
"""Dev tool (GPU): the fp8 block GEMMs at the bench's step-batch shapes, bf16 kernel beside them (same box, same call).
   [OMNI_DEV_LIB=...abl/libomni_<variant>.so] python tools/bench_fp8_gemm.py [iters]
Prints TF/s, the quantise pass of the A operand, and fp8-vs-bf16 rel_l2 of the result (e4m3 operand rounding)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.devlib  # noqa: E402,F401
from vllm_omni_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device=dev).manual_seed(0)
D, R = 3072, 5
Mi, Mt = 2 * R * 4096, 2 * R * 64


def rn(*shape, s=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * s).to(BF16)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, N, K, epi in (("mlp_up", 4 * D, D, ops.EPI_BIAS_GELU_TANH), ("mlp_down", D, 4 * D, ops.EPI_BIAS), ("qkv", 3 * D, D, ops.EPI_BIAS),
                        ("out", D, D, ops.EPI_BIAS)):
    xi, xt, wi, wt, b = rn(Mi, K), rn(Mt, K), rn(N, K, s=0.02), rn(N, K, s=0.02), rn(N)
    xib, xtb, wib, wtb = (ops.w_to_k32_blocked(t) for t in (xi, xt, wi, wt))
    oi, ot = torch.empty(Mi, N, dtype=BF16, device=dev), torch.empty(Mt, N, dtype=BF16, device=dev)
    o8i, o8t = torch.empty_like(oi), torch.empty_like(ot)
    f16 = lambda: ops.gemm([ops.GemmGroupArgs(xib, wib, b, oi, a_k32_blocked=True), ops.GemmGroupArgs(xtb, wtb, b, ot, a_k32_blocked=True)],  # noqa: E731
                           epi, w_k32_blocked=True)
    xi8, xis = ops.quantize_fp8_rows(xib, x_k32_blocked=True)
    xt8, xts = ops.quantize_fp8_rows(xtb, x_k32_blocked=True)
    wi8, wis = ops.quantize_fp8_rows(wi)
    wt8, wts = ops.quantize_fp8_rows(wt)
    f8 = lambda: ops.gemm([ops.GemmGroupArgs(xi8, wi8, b, o8i, a_scale=xis, w_scale=wis, a_k32_blocked=True),  # noqa: E731
                           ops.GemmGroupArgs(xt8, wt8, b, o8t, a_scale=xts, w_scale=wts, a_k32_blocked=True)], epi, fp8=True,
                          w_k32_blocked=True)
    fq = lambda: ops.quantize_fp8_rows(xib, x_k32_blocked=True, out=xi8, scale=xis)  # noqa: E731
    t16, t8, tq = timed(f16), timed(f8), timed(fq)
    fl = 2.0 * (Mi + Mt) * N * K
    rel = float((o8i.float() - oi.float()).norm() / oi.float().norm())
    print(f"{name}: bf16 {fl / t16 / 1e6:7.1f} TF/s ({t16:7.1f} us)  fp8 {fl / t8 / 1e6:7.1f} TF/s ({t8:7.1f} us)  quantize A {tq:6.1f} us "
          f"({Mi * K * 3 / tq / 1e6:.2f} TB/s)  fp8 vs bf16 rel_l2 {rel:.2e}", flush=True)
